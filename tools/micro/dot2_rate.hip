// microbenchmark: issue rate of v_dot2c_f32_bf16 / v_dot2c_f32_f16 / v_cvt_pk_bf16_f32 against v_fma_f32, 8 waves per CU
// (2 per SIMD) of independent accumulator chains.   hipcc --offload-arch=gfx950 -O3 -o dot2_rate dot2_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, const float* w, const unsigned* wp, int iters) {
  float a[16];
  unsigned p[8];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; i++) p[i] = 0x3f803f80u + threadIdx.x + i;     // packed VGPR operands
  const float b = 1.0001f + w[0];
  const unsigned ws = wp[0] + 0x3c003c00u;                              // uniform packed pair -> SGPR
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[(i + 1) & 15], b, a[i]);
        if (MODE == 1) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p[i & 7]), __builtin_bit_cast(bf2, p[(i + 3) & 7]), a[i], false);
        if (MODE == 2) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ws), __builtin_bit_cast(bf2, p[i & 7]), a[i], false);
        if (MODE == 3) a[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, ws), __builtin_bit_cast(h2, p[i & 7]), a[i], false);
        if (MODE == 4) {   // pack two fp32 -> bf16 pair, consumed by a dot2 so it is not dead
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i & 7]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
        }
        if (MODE == 5) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(p[i & 7]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
      }
    if (MODE == 4 || MODE == 5) a[0] += __builtin_bit_cast(float, p[it & 7] & 0x7fff0000u) * 1e-30f;
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += a[i];
  for (int i = 0; i < 8; i++) s += (float)p[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(float* d, float* w, const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, w, (const unsigned*)w, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, w, (const unsigned*)w, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ins_per_lane = (double)iters * 4 * 16;
  printf("%-44s %.3f ms  %.2f instr lanes per CU and ns\n", name, ms, ins_per_lane * 512 / (ms * 1e6));
}
int main() {
  float *d, *w; hipMalloc(&d, 1 << 26); hipMalloc(&w, 64); hipMemset(w, 0, 64);
  run<0>(d, w, "v_fma_f32");
  run<1>(d, w, "v_dot2c_f32_bf16 (VGPR x VGPR)");
  run<2>(d, w, "v_dot2c_f32_bf16 (SGPR x VGPR)");
  run<3>(d, w, "v_dot2c_f32_f16 (SGPR x VGPR)");
  run<4>(d, w, "v_cvt_pk_bf16_f32");
  run<5>(d, w, "v_alignbit_b32");
  return 0;
}
