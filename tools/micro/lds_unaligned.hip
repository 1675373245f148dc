// microbenchmark / probe: do 2-byte-aligned ds_write_b64 / ds_read_b64 work on gfx950 (unaligned access mode), and at what cost?
//   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OFF>
__global__ void k(unsigned* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 48 + 64];
  const int lane = threadIdx.x & 63;
  unsigned char* p = lds + lane * 48 + OFF;           // OFF = 0: aligned, 2: 2-byte aligned
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
    const unsigned long long v = 0x0004000300020001ull + (unsigned long long)lane * 0x0001000100010001ull + it;
    asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(uintptr_t)p), "v"(v) : "memory");
    unsigned long long r;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(uintptr_t)p) : "memory");
    acc += (unsigned)r + (unsigned)(r >> 32) - (unsigned)v - (unsigned)(v >> 32);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;     // 0 if every read returned what was written
}
template <int OFF> void run(unsigned* d, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OFF>, dim3(256), dim3(256), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OFF>, dim3(256), dim3(256), 0, 0, d, 20000);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  unsigned bad = 0; for (int i = 0; i < 256; i++) bad |= h[i];
  printf("%-28s %.3f ms  %s\n", name, ms, bad ? "MISMATCH" : "values ok");
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 1 << 20);
  run<0>(d, "ds b64 aligned");
  run<2>(d, "ds b64 at 2-byte alignment");
  run<4>(d, "ds b64 at 4-byte alignment");
  return 0;
}
