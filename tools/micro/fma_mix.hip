// microbenchmark: VALU issue rate of the fused depthwise backward's FMA mix, no memory, no LDS.
//   MODE 1: 216 v_fma per plane and thread into 27 + 12 accumulators (the dw_pk.hip inner loop: weights in SGPRs) fed by
//           6 ds_read_b128 + 6 ds_read_b64 window reads from a conflict-free LDS image
//   MODE 2: MODE 1 + one s_barrier per plane
// Measured on MI355X (r02): 166-179 / 176-185 lane-FMA per ns and CU = 62-70 % of the VALU peak (128 lanes per clock and
// CU at ~2.0 GHz under this load).  The fused backward kernel runs the same mix plus ~60 % more instructions (staging,
// emit, addressing) at 57 % issue utilisation: its distance from this ceiling is the instruction count, not scheduling.
// 16 waves per CU (4 per SIMD) as 64-thread workgroups.   hipcc --offload-arch=gfx950 -O3 -o fma_mix fma_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64, 4) void k(float* out, const float* w, int planes) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 16 * 16 + 64];
  for (int i = threadIdx.x; i < 2 * 16 * 16 + 64; i += 64) lds[i] = i * 1e-3f;
  __syncthreads();
  float wg[27];
  for (int i = 0; i < 27; i++) wg[i] = w[i];                     // uniform -> SGPRs
  float dW[27], dA[3][4], dB[3][4];
  for (int i = 0; i < 27; i++) dW[i] = 0.f;
  for (int k2 = 0; k2 < 3; k2++) for (int i = 0; i < 4; i++) { dA[k2][i] = 0.f; dB[k2][i] = threadIdx.x * 1e-3f + i + k2; }
  const int r = threadIdx.x >> 2, s = threadIdx.x & 3;
  float winA[3][6], winB[3][6];
  for (int kh = 0; kh < 3; kh++) for (int j = 0; j < 6; j++) { winA[kh][j] = threadIdx.x + kh + j * 0.5f; winB[kh][j] = threadIdx.x - kh + j * 0.25f; }
  for (int t = 0; t < planes; t++) {
    if (MODE >= 1) {
      const float* A = lds + (t & 1) * 256 + (r % 14) * 16 + s * 4;
#pragma unroll
      for (int kh = 0; kh < 3; kh++) {
        const f4 v = *(const f4*)(A + kh * 16);
        const float2 u = *(const float2*)(A + kh * 16 + 4);
        winA[kh][0] = v[0]; winA[kh][1] = v[1]; winA[kh][2] = v[2]; winA[kh][3] = v[3]; winA[kh][4] = u.x; winA[kh][5] = u.y;
      }
#pragma unroll
      for (int kh = 0; kh < 3; kh++) {
        const f4 v = *(const f4*)(A + kh * 16 + 8);
        const float2 u = *(const float2*)(A + kh * 16 + 12);
        winB[kh][0] = v[0]; winB[kh][1] = v[1]; winB[kh][2] = v[2]; winB[kh][3] = v[3]; winB[kh][4] = u.x; winB[kh][5] = u.y;
      }
    }
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
      for (int kw = 0; kw < 3; kw++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float av = winA[kh][i + kw];
          dW[kh * 3 + kw] += dB[0][i] * av;
          dW[9 + kh * 3 + kw] += dB[1][i] * av;
          dW[18 + kh * 3 + kw] += dB[2][i] * av;
        }
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
      for (int kw = 0; kw < 3; kw++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float v = winB[2 - kh][i + 2 - kw];
          dA[0][i] += wg[kh * 3 + kw] * v;
          dA[1][i] += wg[9 + kh * 3 + kw] * v;
          dA[2][i] += wg[18 + kh * 3 + kw] * v;
        }
    if (MODE == 0) {   // keep the inputs changing so nothing is hoisted
      dB[0][0] = dA[0][0] * 1e-6f; winA[0][0] += 1e-6f; winB[1][1] += 1e-6f;
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) dB[0][i] = winB[1][i + 1];
    }
    if (MODE == 2) __syncthreads();
  }
  float sacc = 0.f;
  for (int i = 0; i < 27; i++) sacc += dW[i];
  for (int k2 = 0; k2 < 3; k2++) for (int i = 0; i < 4; i++) sacc += dA[k2][i];
  out[blockIdx.x * 64 + threadIdx.x] = sacc;
}
template <int MODE> void run(float* d, float* w, const char* name) {
  const int planes = 4000, blocks = 256 * 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, w, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, w, planes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fma = (double)planes * 216.0 * 64 * blocks;          // lane-FMAs
  printf("%-40s %.3f ms  %.1f lane-FMA/ns/CU (peak 32 lanes x 4 SIMDs x f GHz)  %.2f ns per wave-plane\n", name, ms,
         fma / (ms * 1e6) / 256, ms * 1e6 / ((double)planes * blocks / 256.0 / 16.0) / 1.0);
}
int main() {
  float *d, *w; hipMalloc(&d, 1 << 24); hipMalloc(&w, 256); hipMemset(w, 0, 256);
  run<1>(d, w, "fma + window reads");
  run<2>(d, w, "fma + window reads + barrier");
  return 0;
}
