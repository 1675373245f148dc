// microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (plain VGPR pairs / broadcast op_sel / SGPR-pair operand)
// with 8 waves per CU (2 per SIMD) of independent chains.   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, const float* w, int iters) {
  v2f a[8];
  for (int i = 0; i < 8; i++) a[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const float b = 1.0001f + w[0], c = 0.5f + w[1];
  const v2f bs = {b, c};       // uniform -> SGPR pair
  v2f bv = {b + threadIdx.x * 1e-9f, c};   // VGPR pair
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) { a[i].x = __builtin_fmaf(a[i].x, bv.x, bv.y); a[i].y = __builtin_fmaf(a[i].y, bv.x, bv.y); }   // 2 scalar FMAs
        if (MODE == 1) a[i] = __builtin_elementwise_fma(a[i], bv, bv);                                                  // packed, VGPR pairs
        if (MODE == 2) a[i] = __builtin_elementwise_fma(a[i], (v2f){bv.x, bv.x}, bv);                                   // packed, broadcast
        if (MODE == 3) a[i] = __builtin_elementwise_fma(bs, (v2f){a[(i + 1) & 7].x, a[(i + 1) & 7].x}, a[i]);          // SGPR pair x broadcast + acc
      }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(float* d, float* w, const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, w, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, w, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fma_per_lane = (double)iters * 4 * 8 * 2;
  printf("%-44s %.3f ms  %.1f G lane-FMA/s per CU  (%.2f FMA lanes per CU and ns)\n", name, ms,
         fma_per_lane * 512 / (ms * 1e-3) * 1e-9, fma_per_lane * 512 / (ms * 1e6));
}
int main() {
  float *d, *w; hipMalloc(&d, 1 << 26); hipMalloc(&w, 64); hipMemset(w, 0, 64);
  run<0>(d, w, "v_fma_f32 x2");
  run<1>(d, w, "v_pk_fma_f32 (VGPR pairs)");
  run<2>(d, w, "v_pk_fma_f32 (broadcast op_sel)");
  run<3>(d, w, "v_pk_fma_f32 (SGPR pair x broadcast VGPR)");
  return 0;
}
