// microbenchmark: what a VALU-bound kernel really gets on MI355X -- instruction rate per SIMD in SHADER CLOCKS (s_memtime)
// and the shader clock itself (s_memtime against the constant 100 MHz s_memrealtime) while the whole chip runs the loop,
// for v_fma_f32 / v_dot2c_f32_bf16 / v_cvt_pk_bf16_f32 and a depthwise-like mix, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_clock valu_clock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, unsigned long long* clk, const float* w, int iters) {
  float a[16];
  unsigned p[8];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; i++) p[i] = 0x3f803f80u + threadIdx.x + i;
  const float b = 1.0001f + w[0];
  const unsigned ws = ((const unsigned*)w)[1] + 0x3c003c00u;
  const unsigned long long c0 = __builtin_readcyclecounter();       // s_memtime
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[(i + 5) & 15], b, a[i]);                       // v_fmac (SGPR x VGPR + acc)
        if (MODE == 1) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ws), __builtin_bit_cast(bf2, p[i & 7]), a[i], false);
        if (MODE == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i & 7]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
        if (MODE == 3) a[i] = __builtin_fmaf(a[(i + 5) & 15], a[(i + 9) & 15], a[i]);          // v_fma, three VGPR operands
        if (MODE == 4) {   // mix: 2 fma : 1 dot2 : 0.5 cvt
          a[i] = __builtin_fmaf(a[(i + 5) & 15], b, a[i]);
          if (i & 1) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ws), __builtin_bit_cast(bf2, p[i & 7]), a[i], false);
          if ((i & 3) == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i & 7]) : "v"(a[(i + 1) & 15]), "v"(a[(i + 2) & 15]));
        }
        if (MODE == 6) a[i] = __builtin_fmaf(a[(i + 1) & 15], a[(i + 2) & 15], a[i]);          // banks i+1, i+2, i
        if (MODE == 7) a[i] = __builtin_fmaf(a[(i + 4) & 15], a[(i + 8) & 15], a[i]);          // all three in one bank (mod 4)
        if (MODE == 8) a[i] = __builtin_fmaf(a[(i + 4) & 15], a[(i + 1) & 15], a[i]);          // src0 and acc in one bank
        if (MODE == 9) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, p[(i + 1) & 7]), __builtin_bit_cast(bf2, p[(i + 2) & 7]), a[i], false);
        if (MODE == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
      }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 16; i++) s += a[i];
  for (int i = 0; i < 8; i++) s += (float)p[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int MODE> void run(float* d, unsigned long long* clk, float* w, const char* name, int waves_per_simd) {
  const int iters = 20000;
  const int threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, clk, w, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, clk, w, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double per_iter = MODE == 4 ? 64.0 * 1.75 : 64.0;
  const double ins = (double)iters * per_iter * waves_per_simd;          // wave-instructions per SIMD
  printf("%-34s %d waves/SIMD  %.3f ms  %.2f shader clocks per wave-instruction and SIMD  shader clock %.0f MHz  (%.1f lanes per CU and ns)\n", name,
         waves_per_simd, ms, (double)h[0] / ins, (double)h[0] / ((double)h[1] / 100.0), ins * 4 * 64 / (ms * 1e6));
}
int main() {
  float *d, *w; unsigned long long* clk;
  (void)hipMalloc(&d, 1 << 26); (void)hipMalloc(&w, 64); (void)hipMemset(w, 0, 64); (void)hipMalloc(&clk, 64);
  for (int wps = 2; wps <= 4; wps *= 2) {
    run<0>(d, clk, w, "v_fmac_f32 (SGPR x VGPR)", wps);
    run<3>(d, clk, w, "v_fma_f32 (3 VGPRs)", wps);
    run<1>(d, clk, w, "v_dot2c_f32_bf16 (SGPR x VGPR)", wps);
    run<2>(d, clk, w, "v_cvt_pk_bf16_f32", wps);
    run<5>(d, clk, w, "v_mov_b32", wps);
    run<4>(d, clk, w, "mix fma : dot2 : cvt = 4 : 2 : 1", wps);
    run<6>(d, clk, w, "v_fma a[i+1]*a[i+2]+a[i]", wps);
    run<7>(d, clk, w, "v_fma a[i+4]*a[i+8]+a[i]", wps);
    run<8>(d, clk, w, "v_fma a[i+4]*a[i+1]+a[i]", wps);
    run<9>(d, clk, w, "v_dot2c p[i+1].p[i+2]+a[i]", wps);
  }
  return 0;
}
