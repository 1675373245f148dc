// Stand-alone C++ microbenchmark of the depthwise kernels through the C ABI (no Python, no torch):
//   hipcc -O2 tools/bench_dw3d.cpp -Iinclude -Lx3d-tf_amd -lx3d_hip -Wl,-rpath,$PWD/x3d-tf_amd -o tools/bench_dw3d
// Prints, for every depthwise shape of X3D-M at batch N (SURVEY 8d "depthwise microbench shapes"), the HIP-event
// time of x3d_dw3d_fwd / x3d_dw3d_bwd, the algorithmic bytes (fwd e*(X + Y), fused bwd e*(X + dY + dX)) and the
// resulting GB/s, plus the kernel instantiation the library dispatches to (x3d_dw3d_kernel_name).
// Inputs are uniform random bf16 bit patterns of small magnitude; results are not checked here (tests/ does that).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "x3d_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define X3D_OK_(x) do { int s_ = (x); if (s_ != X3D_OK) { fprintf(stderr, "%s: %s\n", #x, x3d_last_error()); exit(3); } } while (0)

static uint16_t bf16_bits(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

static void* dev_random_bf16(size_t n, unsigned seed) {
  std::vector<uint16_t> h(n);
  uint32_t s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = bf16_bits(((int)(s >> 8) % 2001 - 1000) * 1e-3f); }
  void* d; HIP_OK(hipMalloc(&d, n * 2)); HIP_OK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}
static float* dev_random_f32(size_t n, unsigned seed, float scale, float offset) {
  std::vector<float> h(n);
  uint32_t s = seed * 40503u + 7u;
  for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = offset + scale * (((int)(s >> 8) % 2001 - 1000) * 1e-3f); }
  float* d; HIP_OK(hipMalloc((void**)&d, n * 4)); HIP_OK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

template <typename F>
static double time_us(F f, int reps) {
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  f(); f();
  HIP_OK(hipDeviceSynchronize());
  std::vector<float> t(reps);
  for (int i = 0; i < reps; i++) {
    HIP_OK(hipEventRecord(e0, 0)); f(); HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipEventElapsedTime(&t[i], e0, e1));
  }
  std::sort(t.begin(), t.end());
  return 1e3 * t[reps / 2];
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64, T = 16;
  struct Shape { int C, H, stride; } shapes[] = {{54, 112, 2}, {54, 56, 1}, {108, 56, 2}, {108, 28, 1},
                                                {216, 28, 2}, {216, 14, 1}, {432, 14, 2}, {432, 7, 1}};
  printf("# libx3d_hip version %d, batch %d, T %d, bf16 storage\n", x3d_version(), N, T);
  printf("%-28s %-44s %10s %10s %10s\n", "shape", "kernel", "us", "MB(alg)", "GB/s(alg)");
  for (const Shape& sh : shapes) {
    const int C = sh.C, H = sh.H, W = sh.H, S = sh.stride, Ho = (H + S - 1) / S, Wo = (W + S - 1) / S;
    const size_t xin = (size_t)N * C * T * H * W, yout = (size_t)N * C * T * Ho * Wo;
    void* x = dev_random_bf16(xin, 1); void* y = dev_random_bf16(yout, 2);
    void* dv = dev_random_bf16(yout, 3); void* ga = dev_random_bf16(xin, 4);
    float* w = dev_random_f32((size_t)C * 27, 5, 0.3f, 0.f);
    float* ss = dev_random_f32((size_t)C * 2, 6, 0.2f, 0.6f);
    float* coef = dev_random_f32((size_t)N * C * 4, 7, 0.3f, 0.5f);
    double *stats, *pool, *asums; float* dw;
    const size_t stats_bytes = (size_t)x3d_stats_replicas() * x3d_stats_stride(C) * 8;   // replicated accumulator
    HIP_OK(hipMalloc((void**)&stats, stats_bytes)); HIP_OK(hipMalloc((void**)&pool, (size_t)N * C * 8));
    HIP_OK(hipMalloc((void**)&asums, C * 16)); HIP_OK(hipMalloc((void**)&dw, C * 27 * 4));
    HIP_OK(hipMemset(stats, 0, stats_bytes)); HIP_OK(hipMemset(pool, 0, (size_t)N * C * 8));
    HIP_OK(hipMemset(asums, 0, C * 16)); HIP_OK(hipMemset(dw, 0, C * 27 * 4));

    x3d_dw3d_fwd_args f; memset(&f, 0, sizeof(f));
    f.x = x; f.w = w; f.y = y; f.in_scale_shift = ss; f.in_act = X3D_ACT_RELU; f.stats = stats; f.pool = pool;
    f.N = N; f.C = C; f.T = T; f.H = H; f.W = W; f.stride = S; f.dtype = X3D_BF16;
    x3d_dw3d_bwd_args b; memset(&b, 0, sizeof(b));
    b.dv = dv; b.braw = y; b.coef_nc = coef; b.araw = x; b.a_scale_shift = ss; b.w = w; b.ga = ga; b.a_sums = asums; b.dw = dw;
    b.N = N; b.C = C; b.T = T; b.H = H; b.W = W; b.stride = S; b.dtype = X3D_BF16;

    char name[128], shp[64];
    snprintf(shp, sizeof(shp), "C%d %dx%dx%d s%d", C, T, H, W, S);
    X3D_OK_(x3d_dw3d_kernel_name(&f, nullptr, name, sizeof(name)));
    const double uf = time_us([&] { X3D_OK_(x3d_dw3d_fwd(&f, nullptr)); }, 15);
    const double bf = 2.0 * (xin + yout);
    printf("%-28s %-44s %10.1f %10.1f %10.1f\n", shp, name, uf, bf / 1e6, bf / uf / 1e3);
    X3D_OK_(x3d_dw3d_kernel_name(nullptr, &b, name, sizeof(name)));
    const double ub = time_us([&] { X3D_OK_(x3d_dw3d_bwd(&b, nullptr)); }, 15);
    const double bb = 2.0 * (2 * xin + yout);
    printf("%-28s %-44s %10.1f %10.1f %10.1f\n", shp, name, ub, bb / 1e6, bb / ub / 1e3);
    void* frees[] = {x, y, dv, ga, w, ss, coef, stats, pool, asums, dw};
    for (void* p : frees) HIP_OK(hipFree(p));
  }
  return 0;
}
