// microbenchmark: VALU issue rate per SIMD with W waves resident (independent v_fma_f32 chains)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, int iters) {
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
  const float b = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 16; i++) a[i] = a[i] * b + c;
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 1 << 26);
  const int iters = 20000;
  for (int wpb : {1, 2, 4, 8}) {           // waves per block; grid = 256 CUs * 4 SIMDs -> blocks = 1024 / ... one block per SIMD
    for (int bps = 1; bps <= 1; bps++) {
      const int threads = 64 * wpb;          // all waves of a block on one CU, spread over 4 SIMDs
      const int blocks = 256;                // one block per CU
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_wave = (double)iters * 64;
      // waves per SIMD = wpb / 4 (rounded up); time per instruction per wave
      printf("waves/CU %2d : %.3f ms, %.2f ns per FMA per wave, wave-instr/s per CU %.3g\n", wpb, ms,
             ms * 1e6 / instr_per_wave, wpb * instr_per_wave / (ms * 1e-3));
    }
  }
  for (int wpb : {4, 8, 16}) {
    const int threads = 256, blocks = 256 * (wpb / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64;
    printf("256-thread blocks, waves/CU %2d : %.3f ms, wave-instr/s per CU %.3g\n", wpb, ms, wpb * instr_per_wave / (ms * 1e-3));
  }
  return 0;
}
