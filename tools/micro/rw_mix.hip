// microbenchmark: what does HBM give a STREAM whose reads and writes are not 1 : 1?  The write-heavy pointwise convs of the
// step (24 -> 54 @ 112^2 / 56^2, 48 -> 108 @ 28^2: output 2.25 x the input) run at 2.7 - 3.1 TB/s of bytes moved, the read-heavy
// `c` convs on the same planes at 4.2 - 4.4: is that the kernels or the mix?  A bare kernel -- 16 bytes per lane, fully
// coalesced, NR vectors read and NW vectors written per iteration, no arithmetic beyond keeping the loads alive -- at
// read : write = 4 : 9 (1 : 2.25), 1 : 1, 9 : 4 and the pure forms, at 4 / 8 / 16 waves per CU, plain and non-temporal stores.
// Each arm moves ~2 GB (far past the 256 MB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -o rw_mix rw_mix.hip && ./rw_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void rw_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long G, int iters) {
  const long long gtid = (long long)blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; it++) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    u32x4 r[NR > 0 ? NR : 1];
#pragma unroll
    for (int j = 0; j < NR; j++) r[j] = src[((long long)it * NR + j) * G + gtid];
#pragma unroll
    for (int j = 0; j < NR; j++) acc ^= r[j];
#pragma unroll
    for (int j = 0; j < NW; j++) {
      u32x4 v = acc;
      v[0] += (unsigned)j;
      u32x4* p = &dst[((long long)it * NW + j) * G + gtid];
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
    if (NW == 0 && acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) dst[gtid] = acc;   // (keeps the loads of the read-only arm alive)
  }
}

template <int NR, int NW, bool NT>
static void run(const u32x4* src, u32x4* dst, int cus, int wpc, const char* name) {
  const int grid = cus * wpc / 4;                    // 256-thread workgroups: wpc waves per CU resident
  const long long G = (long long)grid * 256;
  const double target = 2.0e9;                        // bytes per launch
  int iters = (int)(target / ((double)(NR + NW) * 16.0 * (double)G));
  if (iters < 1) iters = 1;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((rw_kernel<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, src, dst, G, iters);
  (void)hipDeviceSynchronize();
  const int reps = 5;
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((rw_kernel<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, src, dst, G, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)(NR + NW) * 16.0 * (double)G * iters;
  printf("%-22s %2d waves/CU  %7.1f MB read %7.1f MB written  %7.1f us  %5.2f TB/s\n", name, wpc, NR * 16.0 * G * iters / 1e6,
         NW * 16.0 * G * iters / 1e6, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
}

int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  u32x4 *src, *dst;
  const size_t cap = (size_t)2200 << 20;
  (void)hipMalloc(&src, cap); (void)hipMalloc(&dst, cap);
  (void)hipMemset(src, 1, cap); (void)hipMemset(dst, 0, cap);
  const int wpcs[3] = {4, 8, 16};
  for (int k = 0; k < 3; k++) {
    const int w = wpcs[k];
    run<1, 0, false>(src, dst, cus, w, "read only");
    run<0, 1, false>(src, dst, cus, w, "write only");
    run<0, 1, true>(src, dst, cus, w, "write only, nt");
    run<1, 1, false>(src, dst, cus, w, "1 : 1");
    run<1, 1, true>(src, dst, cus, w, "1 : 1, nt stores");
    run<4, 9, false>(src, dst, cus, w, "1 : 2.25");
    run<4, 9, true>(src, dst, cus, w, "1 : 2.25, nt stores");
    run<9, 4, false>(src, dst, cus, w, "2.25 : 1");
    run<9, 4, true>(src, dst, cus, w, "2.25 : 1, nt stores");
  }
  return 0;
}
