// BatchNorm bookkeeping and the bandwidth-bound elementwise / reduction kernels around the convs:
// BN finalize (train / eval / backward), residual tail (Add+ReLU) forward and backward, ReLU+BN
// backward reduce (stem, conv5 after global pooling), pool5, NTHWC<->NCTHW at the module boundary.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// BN finalize kernels: one thread per channel (C <= a few hundred)
// ------------------------------------------------------------------------------------------------
// One lane per (channel, copy): 32 lanes load the 32 copies of a channel in ONE round trip and add them with an xor
// butterfly (1, 2, 4, 8, 16); a 64-thread block finalizes two channels.  (One thread per channel summing the copies
// itself is four dependent round trips: 84 such launches sit between producers and consumers in every train step.)
static_assert(STATS_R == 32, "bn_finalize_kernel maps one half-wave to the copies of a channel");
__global__ __launch_bounds__(64) void bn_finalize_kernel(const x3d_bn_fold f, int C) {
  const int c = blockIdx.x * 2 + (threadIdx.x >> 5), r = threadIdx.x & 31;
  const int cc = c < C ? c : C - 1;
  const long long rs = stats_stride(C);
  double s1 = f.stats[r * rs + cc * 2], s2 = f.stats[r * rs + cc * 2 + 1];
  const float ga = f.gamma[cc], be = f.beta[cc];
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  float sc, sh;
  bn_coefs(f, cc, s1, s2, ga, be, r == 0 && c < C, sc, sh);
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, double count,
                                       const float* __restrict__ mi, const float* __restrict__ gamma,
                                       float* coef, float* dgamma, float* dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float A, B, Cc;
  double dga, dbe;
  bn_bwd_coefs(sums, count, mi, gamma, c, A, B, Cc, dga, dbe);     // (common.h: the one definition)
  coef[c * 4] = A;
  coef[c * 4 + 1] = B;
  coef[c * 4 + 2] = Cc;
  coef[c * 4 + 3] = 0.f;
  dgamma[c] += (float)dga;
  dbeta[c] += (float)dbe;
}

extern "C" int x3d_stats_replicas(void) { return STATS_R; }
extern "C" long long x3d_stats_stride(int C) { return stats_stride(C); }

extern "C" int x3d_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                               float* moving_mean, float* moving_var, float eps, float momentum,
                               int update_moving, float* scale_shift, float* mean_invstd, int C,
                               void* stream) {
  X3D_REQUIRE(stats && gamma && beta && scale_shift && mean_invstd && C > 0 && count > 0, "bn_finalize: bad args");
  X3D_REQUIRE(!update_moving || (moving_mean && moving_var), "bn_finalize: moving stats required");
  x3d_bn_fold f;
  f.stats = stats; f.count = count; f.gamma = gamma; f.beta = beta; f.moving_mean = moving_mean; f.moving_var = moving_var;
  f.eps = eps; f.momentum = momentum; f.update_moving = update_moving; f.scale_shift = scale_shift; f.mean_invstd = mean_invstd;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 2)), dim3(64), 0, (hipStream_t)stream, f, C);
  X3D_LAUNCH_CHECK("bn_finalize");
  return X3D_OK;
}

__global__ __launch_bounds__(64) void bn_eval_coef_batched_kernel(const x3d_bn_eval_item* __restrict__ items, float eps) {
  const x3d_bn_eval_item it = items[blockIdx.x];
  for (int c = threadIdx.x; c < it.C; c += 64) {
    const float invstd = 1.0f / sqrtf(it.moving_var[c] + eps);
    const float sc = it.gamma[c] * invstd;
    it.scale_shift[c * 2] = sc;
    it.scale_shift[c * 2 + 1] = it.beta[c] - it.moving_mean[c] * sc;
    it.mean_invstd[c * 2] = it.moving_mean[c];
    it.mean_invstd[c * 2 + 1] = invstd;
  }
}

extern "C" int x3d_bn_eval_coef_batched(const x3d_bn_eval_item* items, int n_items, float eps, void* stream) {
  X3D_REQUIRE(items && n_items > 0, "bn_eval_coef_batched: no items");
  hipLaunchKernelGGL(bn_eval_coef_batched_kernel, dim3(n_items), dim3(64), 0, (hipStream_t)stream, items, eps);
  X3D_LAUNCH_CHECK("bn_eval_coef_batched");
  return X3D_OK;
}

extern "C" int x3d_bn_bwd_finalize(const double* sums, double count, const float* mean_invstd,
                                   const float* gamma, float* coef, float* dgamma, float* dbeta, int C,
                                   void* stream) {
  X3D_REQUIRE(sums && mean_invstd && gamma && coef && dgamma && dbeta && C > 0 && count > 0, "bn_bwd_finalize: bad args");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, count,
                     mean_invstd, gamma, coef, dgamma, dbeta, C);
  X3D_LAUNCH_CHECK("bn_bwd_finalize");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// plane-wise elementwise kernels.  grid = (chunks over P, N*C): the channel is uniform per block.
// Each thread handles VEC contiguous elements per iteration.
// ------------------------------------------------------------------------------------------------
#define ELEM_BLOCK 256
#define ELEM_ITERS 4

static inline dim3 elem_grid(long long P, int vec, int NC) {
  return dim3((unsigned)ceil_div_ll(P, (long long)ELEM_BLOCK * vec * ELEM_ITERS), (unsigned)NC);
}

// FOLD: the BatchNorm finalize of bn_c / bn_r runs here (x3d_bn_fold); the first workgroup of each channel of
// sample 0 publishes the coefficients and updates the moving statistics
struct TailFold { x3d_bn_fold c, r; int has_r; };
template <typename T, int VEC, bool FOLD>
__global__ __launch_bounds__(ELEM_BLOCK) void tail_fwd_kernel(const T* __restrict__ craw,
                                                             const float* __restrict__ ssc,
                                                             const T* __restrict__ sh,
                                                             const float* __restrict__ ssr, T* __restrict__ y,
                                                             int C, long long P, const TailFold fold) {
  const int nc = blockIdx.y, c = nc % C;
  float sc, tc, sr = 1.f, tr = 0.f;
  if constexpr (FOLD) {
    const bool writer = (nc < C) && blockIdx.x == 0 && threadIdx.x == 0;
    bn_fold_channel(fold.c, c, writer, sc, tc, C);
    if (fold.has_r) bn_fold_channel(fold.r, c, writer, sr, tr, C);
  } else {
    sc = ssc[c * 2]; tc = ssc[c * 2 + 1];
    if (ssr) { sr = ssr[c * 2]; tr = ssr[c * 2 + 1]; }
  }
  const long long base = (long long)nc * P;
  long long p = ((long long)blockIdx.x * ELEM_ITERS * ELEM_BLOCK + threadIdx.x) * VEC;
#pragma unroll
  for (int it = 0; it < ELEM_ITERS; it++, p += (long long)ELEM_BLOCK * VEC) {
    if (p < P) {
      float a[VEC], b[VEC], o[VEC];
      VecIO<T, VEC>::load(craw + base + p, a);
      if (sh) {
        VecIO<T, VEC>::load(sh + base + p, b);
#pragma unroll
        for (int e = 0; e < VEC; e++) o[e] = fmaxf(sc * a[e] + tc + (sr * b[e] + tr), 0.f);
      } else {  // plain BN + ReLU (stem output)
#pragma unroll
        for (int e = 0; e < VEC; e++) o[e] = fmaxf(sc * a[e] + tc, 0.f);
      }
      VecIO<T, VEC>::store(y + base + p, o);
    }
  }
}

// g = dy*[y>0] in place; sums_c += (sum g, sum g*craw); sums_r += (sum g, sum g*rraw)
template <typename T, int VEC>
__global__ __launch_bounds__(ELEM_BLOCK) void tail_bwd_kernel(T* __restrict__ dyg, const T* __restrict__ y,
                                                             const T* __restrict__ craw,
                                                             const T* __restrict__ rraw, double* sums_c,
                                                             double* sums_r, int C, long long P) {
  __shared__ float scratch[3 * (ELEM_BLOCK / 64)];
  const int nc = blockIdx.y, c = nc % C;
  const long long base = (long long)nc * P;
  float red[3] = {0.f, 0.f, 0.f};
  long long p = ((long long)blockIdx.x * ELEM_ITERS * ELEM_BLOCK + threadIdx.x) * VEC;
#pragma unroll
  for (int it = 0; it < ELEM_ITERS; it++, p += (long long)ELEM_BLOCK * VEC) {
    if (p < P) {
      float d[VEC], yy[VEC], cr[VEC], rr[VEC], g[VEC];
      VecIO<T, VEC>::load(dyg + base + p, d);
      VecIO<T, VEC>::load(y + base + p, yy);
      VecIO<T, VEC>::load(craw + base + p, cr);
      if (rraw) VecIO<T, VEC>::load(rraw + base + p, rr);
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        g[e] = yy[e] > 0.f ? d[e] : 0.f;
        red[0] += g[e];
        red[1] += g[e] * cr[e];
        if (rraw) red[2] += g[e] * rr[e];
      }
      VecIO<T, VEC>::store(dyg + base + p, g);
    }
  }
  block_sum<3>(red, scratch);
  if (threadIdx.x == 0) {
    atomic_add_d(&sums_c[c * 2], (double)red[0]);
    atomic_add_d(&sums_c[c * 2 + 1], (double)red[1]);
    if (rraw) {
      atomic_add_d(&sums_r[c * 2], (double)red[0]);
      atomic_add_d(&sums_r[c * 2 + 1], (double)red[2]);
    }
  }
}

// ... small planes (P / VEC < 256: the 7 x 7 stage): one workgroup takes NB samples of ONE channel -- with a workgroup per (n, c)
// 98 of 256 threads had a vector and N workgroups per channel met in the fp64 atomics (192 ch x 64 clips of 16x7x7: 28 us for
// 77 MB).  grid = (C, ceil(N / NB)); the flat index runs over (sample in the group, vector of the plane).
template <typename T, int VEC>
__global__ __launch_bounds__(ELEM_BLOCK) void tail_bwd_small_kernel(T* __restrict__ dyg, const T* __restrict__ y,
                                                                   const T* __restrict__ craw, const T* __restrict__ rraw,
                                                                   double* sums_c, double* sums_r, int N, int C, int P, int NB) {
  __shared__ float scratch[3 * (ELEM_BLOCK / 64)];
  const int c = blockIdx.x, n0 = blockIdx.y * NB;
  const int nb = min(NB, N - n0);
  const int vpp = P / VEC, total = nb * vpp;
  float red[3] = {0.f, 0.f, 0.f};
  for (int i0 = threadIdx.x; i0 < total; i0 += 2 * ELEM_BLOCK) {
    // two vectors per thread and round, their loads issued together
    float d[2][VEC], yy[2][VEC], cr[2][VEC], rr[2][VEC];
    long long off[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int i = i0 + u * ELEM_BLOCK;
      ok[u] = i < total;
      const int nl = ok[u] ? i / vpp : 0, pv = ok[u] ? i - nl * vpp : 0;
      off[u] = ((long long)(n0 + nl) * C + c) * P + (long long)pv * VEC;
      if (ok[u]) {
        VecIO<T, VEC>::load(dyg + off[u], d[u]);
        VecIO<T, VEC>::load(y + off[u], yy[u]);
        VecIO<T, VEC>::load(craw + off[u], cr[u]);
        if (rraw) VecIO<T, VEC>::load(rraw + off[u], rr[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!ok[u]) continue;
      float g[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        g[e] = yy[u][e] > 0.f ? d[u][e] : 0.f;
        red[0] += g[e];
        red[1] += g[e] * cr[u][e];
        if (rraw) red[2] += g[e] * rr[u][e];
      }
      VecIO<T, VEC>::store(dyg + off[u], g);
    }
  }
  block_sum<3>(red, scratch);
  if (threadIdx.x == 0) {
    atomic_add_d(&sums_c[c * 2], (double)red[0]);
    atomic_add_d(&sums_c[c * 2 + 1], (double)red[1]);
    if (rraw) {
      atomic_add_d(&sums_r[c * 2], (double)red[0]);
      atomic_add_d(&sums_r[c * 2 + 1], (double)red[2]);
    }
  }
}

// z = s*yraw + t ; g = (dy ? dy : dpool[n][c]/P) * [z > 0]; sums += (sum g, sum g*yraw)
template <typename T, int VEC>
__global__ __launch_bounds__(ELEM_BLOCK) void relu_bn_bwd_reduce_kernel(const T* __restrict__ dy,
                                                                       const float* __restrict__ dpool,
                                                                       const T* __restrict__ yraw,
                                                                       const float* __restrict__ ss, T* g,
                                                                       double* sums, int C, long long P) {
  __shared__ float scratch[2 * (ELEM_BLOCK / 64)];
  const int nc = blockIdx.y, c = nc % C;
  const float s = ss[c * 2], t = ss[c * 2 + 1];
  const float dp = dpool ? dpool[nc] / (float)P : 0.f;
  const long long base = (long long)nc * P;
  float red[2] = {0.f, 0.f};
  long long p = ((long long)blockIdx.x * ELEM_ITERS * ELEM_BLOCK + threadIdx.x) * VEC;
#pragma unroll
  for (int it = 0; it < ELEM_ITERS; it++, p += (long long)ELEM_BLOCK * VEC) {
    if (p < P) {
      float d[VEC], yr[VEC], o[VEC];
      VecIO<T, VEC>::load(yraw + base + p, yr);
      if (dy) {
        VecIO<T, VEC>::load(dy + base + p, d);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) d[e] = dp;
      }
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        o[e] = (s * yr[e] + t > 0.f) ? d[e] : 0.f;
        const float gr = round_to<T>(o[e]);
        red[0] += gr;
        red[1] += gr * yr[e];
      }
      if (g) VecIO<T, VEC>::store(g + base + p, o);   // g == NULL: reduce only (the consumer applies the mask itself)
    }
  }
  block_sum<2>(red, scratch);
  if (threadIdx.x == 0) {
    atomic_add_d(&sums[c * 2], (double)red[0]);
    atomic_add_d(&sums[c * 2 + 1], (double)red[1]);
  }
}

// pooled[n][c] = mean_p relu(s*x + t): one block per (n, c)
template <typename T, int VEC>
__global__ __launch_bounds__(ELEM_BLOCK) void pool_fwd_kernel(const T* __restrict__ x,
                                                             const float* __restrict__ ss, float* pooled,
                                                             int C, long long P) {
  __shared__ float scratch[ELEM_BLOCK / 64];
  const int nc = blockIdx.x, c = nc % C;
  const float s = ss[c * 2], t = ss[c * 2 + 1];
  const long long base = (long long)nc * P;
  float red[1] = {0.f};
  for (long long p = (long long)threadIdx.x * VEC; p < P; p += (long long)ELEM_BLOCK * VEC) {
    float v[VEC];
    VecIO<T, VEC>::load(x + base + p, v);
#pragma unroll
    for (int e = 0; e < VEC; e++) red[0] += fmaxf(s * v[e] + t, 0.f);
  }
  block_sum<1>(red, scratch);
  if (threadIdx.x == 0) pooled[nc] = red[0] / (float)P;
}

static inline int norm_vec(int dtype, int vec) {
  const int full = dtype == X3D_F32 ? 4 : 8;
  return vec >= full ? full : 1;
}

extern "C" int x3d_tail_fwd(const void* c_raw, const float* c_scale_shift, const void* shortcut,
                            const float* r_scale_shift, void* y, int N, int C, long long P, int dtype,
                            void* stream) {
  X3D_REQUIRE(c_raw && c_scale_shift && y && N > 0 && C > 0 && P > 0, "tail_fwd: bad args");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "tail_fwd: bad dtype");
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const int vec = norm_vec(dtype, pick_vec(eb, P, c_raw, shortcut, y));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = elem_grid(P, vec, N * C);
  TailFold nofold;
  memset(&nofold, 0, sizeof(nofold));
#define ARGS(T) (const T*)c_raw, c_scale_shift, (const T*)shortcut, r_scale_shift, (T*)y, C, P, nofold
  if (dtype == X3D_F32) {
    if (vec == 4) hipLaunchKernelGGL((tail_fwd_kernel<float, 4, false>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
    else hipLaunchKernelGGL((tail_fwd_kernel<float, 1, false>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
  } else if (dtype == X3D_F16) {
    if (vec == 8) hipLaunchKernelGGL((tail_fwd_kernel<f16, 8, false>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
    else hipLaunchKernelGGL((tail_fwd_kernel<f16, 1, false>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
  } else {
    if (vec == 8) hipLaunchKernelGGL((tail_fwd_kernel<bf16, 8, false>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
    else hipLaunchKernelGGL((tail_fwd_kernel<bf16, 1, false>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
  }
#undef ARGS
  X3D_LAUNCH_CHECK("tail_fwd");
  return X3D_OK;
}

extern "C" int x3d_tail_fwd_bn(const void* c_raw, const x3d_bn_fold* c_bn, const void* shortcut, const x3d_bn_fold* r_bn,
                               void* y, int N, int C, long long P, int dtype, void* stream) {
  X3D_REQUIRE(c_raw && y && N > 0 && C > 0 && P > 0, "tail_fwd_bn: bad args");
  X3D_REQUIRE(bn_fold_valid(c_bn), "tail_fwd_bn: incomplete x3d_bn_fold for bn_c");
  X3D_REQUIRE(!r_bn || (shortcut && bn_fold_valid(r_bn)), "tail_fwd_bn: incomplete x3d_bn_fold for bn_r");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "tail_fwd_bn: bad dtype");
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const int vec = norm_vec(dtype, pick_vec(eb, P, c_raw, shortcut, y));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = elem_grid(P, vec, N * C);
  TailFold fold;
  memset(&fold, 0, sizeof(fold));
  fold.c = *c_bn;
  if (r_bn) { fold.r = *r_bn; fold.has_r = 1; }
#define ARGS(T) (const T*)c_raw, (const float*)nullptr, (const T*)shortcut, (const float*)nullptr, (T*)y, C, P, fold
  if (dtype == X3D_F32) {
    if (vec == 4) hipLaunchKernelGGL((tail_fwd_kernel<float, 4, true>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
    else hipLaunchKernelGGL((tail_fwd_kernel<float, 1, true>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
  } else if (dtype == X3D_F16) {
    if (vec == 8) hipLaunchKernelGGL((tail_fwd_kernel<f16, 8, true>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
    else hipLaunchKernelGGL((tail_fwd_kernel<f16, 1, true>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
  } else {
    if (vec == 8) hipLaunchKernelGGL((tail_fwd_kernel<bf16, 8, true>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
    else hipLaunchKernelGGL((tail_fwd_kernel<bf16, 1, true>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
  }
#undef ARGS
  X3D_LAUNCH_CHECK("tail_fwd_bn");
  return X3D_OK;
}

extern "C" int x3d_tail_bwd(void* dy_g, const void* y, const void* c_raw, const void* r_raw, double* sums_c,
                            double* sums_r, int N, int C, long long P, int dtype, void* stream) {
  X3D_REQUIRE(dy_g && y && c_raw && sums_c && N > 0 && C > 0 && P > 0, "tail_bwd: bad args");
  X3D_REQUIRE(!r_raw || sums_r, "tail_bwd: sums_r required with r_raw");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "tail_bwd: bad dtype");
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const int vec = norm_vec(dtype, pick_vec(eb, P, dy_g, y, c_raw, r_raw));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = elem_grid(P, vec, N * C);
  // small planes in 16-bit storage: NB samples of one channel per workgroup (X3D_TAIL_SMALL=0: A/B hook)
  if (dtype != X3D_F32 && vec == 8 && P / 8 < ELEM_BLOCK && x3d_env_int("X3D_TAIL_SMALL", 1) != 0) {
    int nb = (int)(4 * ELEM_BLOCK / (P / 8));
    if (nb > N) nb = N;
    if (nb > 16) nb = 16;
    const dim3 g2((unsigned)C, (unsigned)ceil_div(N, nb));
    if (dtype == X3D_F16)
      hipLaunchKernelGGL((tail_bwd_small_kernel<f16, 8>), g2, dim3(ELEM_BLOCK), 0, st, (f16*)dy_g, (const f16*)y, (const f16*)c_raw,
                         (const f16*)r_raw, sums_c, sums_r, N, C, (int)P, nb);
    else
      hipLaunchKernelGGL((tail_bwd_small_kernel<bf16, 8>), g2, dim3(ELEM_BLOCK), 0, st, (bf16*)dy_g, (const bf16*)y, (const bf16*)c_raw,
                         (const bf16*)r_raw, sums_c, sums_r, N, C, (int)P, nb);
    X3D_LAUNCH_CHECK("tail_bwd");
    return X3D_OK;
  }
#define ARGS(T) (T*)dy_g, (const T*)y, (const T*)c_raw, (const T*)r_raw, sums_c, sums_r, C, P
  if (dtype == X3D_F32) {
    if (vec == 4) hipLaunchKernelGGL((tail_bwd_kernel<float, 4>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
    else hipLaunchKernelGGL((tail_bwd_kernel<float, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
  } else if (dtype == X3D_F16) {
    if (vec == 8) hipLaunchKernelGGL((tail_bwd_kernel<f16, 8>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
    else hipLaunchKernelGGL((tail_bwd_kernel<f16, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
  } else {
    if (vec == 8) hipLaunchKernelGGL((tail_bwd_kernel<bf16, 8>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
    else hipLaunchKernelGGL((tail_bwd_kernel<bf16, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
  }
#undef ARGS
  X3D_LAUNCH_CHECK("tail_bwd");
  return X3D_OK;
}

extern "C" int x3d_relu_bn_bwd_reduce(const void* dy, const float* dpool, const void* yraw,
                                      const float* scale_shift, void* g, double* sums, int N, int C,
                                      long long P, int dtype, void* stream) {
  X3D_REQUIRE((dy != nullptr) != (dpool != nullptr), "relu_bn_bwd_reduce: exactly one of dy / dpool");
  X3D_REQUIRE(yraw && scale_shift && sums && N > 0 && C > 0 && P > 0, "relu_bn_bwd_reduce: bad args");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "relu_bn_bwd_reduce: bad dtype");
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const int vec = norm_vec(dtype, pick_vec(eb, P, dy, yraw, g));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid = elem_grid(P, vec, N * C);
#define ARGS(T) (const T*)dy, dpool, (const T*)yraw, scale_shift, (T*)g, sums, C, P
  if (dtype == X3D_F32) {
    if (vec == 4) hipLaunchKernelGGL((relu_bn_bwd_reduce_kernel<float, 4>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
    else hipLaunchKernelGGL((relu_bn_bwd_reduce_kernel<float, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
  } else if (dtype == X3D_F16) {
    if (vec == 8) hipLaunchKernelGGL((relu_bn_bwd_reduce_kernel<f16, 8>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
    else hipLaunchKernelGGL((relu_bn_bwd_reduce_kernel<f16, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
  } else {
    if (vec == 8) hipLaunchKernelGGL((relu_bn_bwd_reduce_kernel<bf16, 8>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
    else hipLaunchKernelGGL((relu_bn_bwd_reduce_kernel<bf16, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
  }
#undef ARGS
  X3D_LAUNCH_CHECK("relu_bn_bwd_reduce");
  return X3D_OK;
}

extern "C" int x3d_pool_fwd(const void* x_raw, const float* scale_shift, float* pooled, int N, int C,
                            long long P, int dtype, void* stream) {
  X3D_REQUIRE(x_raw && scale_shift && pooled && N > 0 && C > 0 && P > 0, "pool_fwd: bad args");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "pool_fwd: bad dtype");
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const int vec = norm_vec(dtype, pick_vec(eb, P, x_raw));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(N * C));
#define ARGS(T) (const T*)x_raw, scale_shift, pooled, C, P
  if (dtype == X3D_F32) {
    if (vec == 4) hipLaunchKernelGGL((pool_fwd_kernel<float, 4>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
    else hipLaunchKernelGGL((pool_fwd_kernel<float, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(float));
  } else if (dtype == X3D_F16) {
    if (vec == 8) hipLaunchKernelGGL((pool_fwd_kernel<f16, 8>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
    else hipLaunchKernelGGL((pool_fwd_kernel<f16, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(f16));
  } else {
    if (vec == 8) hipLaunchKernelGGL((pool_fwd_kernel<bf16, 8>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
    else hipLaunchKernelGGL((pool_fwd_kernel<bf16, 1>), grid, dim3(ELEM_BLOCK), 0, st, ARGS(bf16));
  }
#undef ARGS
  X3D_LAUNCH_CHECK("pool_fwd");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// module boundary: NTHWC (reference) -> NCTHW (internal).  C is tiny (3), P is huge: one thread per
// point reads C contiguous values and writes C planes (coalesced along P).
// ------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void nthwc_to_ncthw_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int C, long long P) {
  const int n = blockIdx.y;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const TS* s = src + ((long long)n * P + p) * C;
  for (int c = 0; c < C; c++) dst[((long long)n * C + c) * P + p] = from_f<TD>(to_f<TS>(s[c]));
}

// C = 3, P % 8 == 0, 16-byte aligned tensors (the clips of every X3D configuration): a thread takes 8 points -- 24
// contiguous source values as three vector loads -- and writes one 8-point vector per channel plane.  The one-point form
// above moves 2-byte pieces and is instruction bound (190 us for 64 clips of 16 x 224 x 224 against ~40 us of HBM time).
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nthwc_to_ncthw_c3_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long P8) {
  const int n = blockIdx.y;
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // group of 8 points
  if (q >= P8) return;
  float v[3][8];
  const TS* s = src + ((long long)n * P8 + q) * 24;
#pragma unroll
  for (int k = 0; k < 3; k++) VecIO<TS, 8>::load(s + 8 * k, v[k]);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = v[(3 * j + c) >> 3][(3 * j + c) & 7];
    VecIO<TD, 8>::store(dst + ((long long)n * 3 + c) * (P8 * 8) + q * 8, o);
  }
}
template <typename TS, typename TD>
static bool nthwc_c3_fast(const void* src, void* dst, int N, int C, long long P, hipStream_t st) {
  if (C != 3 || (P & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return false;
  const long long P8 = P >> 3;
  dim3 grid((unsigned)ceil_div_ll(P8, 256), (unsigned)N);
  hipLaunchKernelGGL((nthwc_to_ncthw_c3_kernel<TS, TD>), grid, dim3(256), 0, st, (const TS*)src, (TD*)dst, P8);
  return true;
}

extern "C" int x3d_nthwc_to_ncthw(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C,
                                  long long P, void* stream) {
  X3D_REQUIRE(src && dst && N > 0 && C > 0 && P > 0, "nthwc_to_ncthw: bad args");
  dim3 grid((unsigned)ceil_div_ll(P, 256), (unsigned)N);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == X3D_F32 && dst_dtype == X3D_F32)
    { if (!nthwc_c3_fast<float, float>(src, dst, N, C, P, st))
      hipLaunchKernelGGL((nthwc_to_ncthw_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, C, P); }
  else if (src_dtype == X3D_F32 && dst_dtype == X3D_BF16)
    { if (!nthwc_c3_fast<float, bf16>(src, dst, N, C, P, st))
      hipLaunchKernelGGL((nthwc_to_ncthw_kernel<float, bf16>), grid, dim3(256), 0, st, (const float*)src, (bf16*)dst, C, P); }
  else if (src_dtype == X3D_BF16 && dst_dtype == X3D_BF16)
    { if (!nthwc_c3_fast<bf16, bf16>(src, dst, N, C, P, st))
      hipLaunchKernelGGL((nthwc_to_ncthw_kernel<bf16, bf16>), grid, dim3(256), 0, st, (const bf16*)src, (bf16*)dst, C, P); }
  else if (src_dtype == X3D_BF16 && dst_dtype == X3D_F32)
    { if (!nthwc_c3_fast<bf16, float>(src, dst, N, C, P, st))
      hipLaunchKernelGGL((nthwc_to_ncthw_kernel<bf16, float>), grid, dim3(256), 0, st, (const bf16*)src, (float*)dst, C, P); }
  else if (src_dtype == X3D_F32 && dst_dtype == X3D_F16)
    { if (!nthwc_c3_fast<float, f16>(src, dst, N, C, P, st))
      hipLaunchKernelGGL((nthwc_to_ncthw_kernel<float, f16>), grid, dim3(256), 0, st, (const float*)src, (f16*)dst, C, P); }
  else if (src_dtype == X3D_F16 && dst_dtype == X3D_F16)
    { if (!nthwc_c3_fast<f16, f16>(src, dst, N, C, P, st))
      hipLaunchKernelGGL((nthwc_to_ncthw_kernel<f16, f16>), grid, dim3(256), 0, st, (const f16*)src, (f16*)dst, C, P); }
  else if (src_dtype == X3D_F16 && dst_dtype == X3D_F32)
    { if (!nthwc_c3_fast<f16, float>(src, dst, N, C, P, st))
      hipLaunchKernelGGL((nthwc_to_ncthw_kernel<f16, float>), grid, dim3(256), 0, st, (const f16*)src, (float*)dst, C, P); }
  else {
    x3d_set_error("nthwc_to_ncthw: bad dtype");
    return X3D_ERR_INVALID;
  }
  X3D_LAUNCH_CHECK("nthwc_to_ncthw");
  return X3D_OK;
}


// ------------------------------------------------------------------------------------------------------------------------
// Even-pixel copy of a block input: dst[plane][ho][wo] = src[plane][2 ho][2 wo], Ho = ceil(H / 2), Wo = ceil(W / 2) -- the pixels
// a stride-(1,2,2) 'valid' 1x1x1 shortcut conv samples (reference model.py:360-367).  Round 6: with this compact tensor the
// shortcut conv's forward, data gradient and weight gradient are plain dense launches (16-byte coalesced rows) instead of the
// strided gathers, whose rows of 7 / 14 outputs took one output per 4-byte load (96 -> 192 @14^2: 68 us for 29 MB).
// One thread = VEC consecutive outputs of an output row: 2 * VEC input elements in one aligned load (VEC = 8 / 4 / 2), or
// element loads (odd widths).
// ------------------------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void subsample2_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int Ho, int Wo,
                                                         long long rows /* planes * Ho */) {
  const int vpr = (Wo + VEC - 1) / VEC;                       // vectors per output row (VEC > 1: Wo % VEC == 0, host check)
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= rows * vpr) return;
  const long long row = v / vpr;
  const int wo0 = (int)(v - row * vpr) * VEC;
  const long long plane = row / Ho;
  const int ho = (int)(row - plane * Ho);
  const T* s = src + (plane * H + 2 * ho) * (long long)W + 2 * wo0;
  T* d = dst + row * Wo + wo0;
  if constexpr (VEC == 1) {
    d[0] = s[0];
  } else {
    typedef __attribute__((ext_vector_type(2 * VEC))) T in_t;
    typedef __attribute__((ext_vector_type(VEC))) T out_t;
    const in_t a = *(const in_t*)s;
    out_t o;
#pragma unroll
    for (int e = 0; e < VEC; e++) o[e] = a[2 * e];
    *(out_t*)d = o;
  }
}

template <typename T>
static void subsample2_launch(const void* src, void* dst, long long planes, int H, int W, hipStream_t st) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long rows = planes * Ho;
  int vec = 1;
  // 2 * vec input elements per load: rows of whole loads (W % (2 vec) == 0), both tensors aligned for their vector
  for (int v = 16 / (int)sizeof(T); v >= 2; v >>= 1)
    if ((W % (2 * v)) == 0 && ((uintptr_t)src % (2 * v * sizeof(T))) == 0 && ((uintptr_t)dst % (v * sizeof(T))) == 0) { vec = v; break; }
  const long long nv = rows * ((Wo + vec - 1) / vec);
  const unsigned grid = (unsigned)ceil_div_ll(nv, 256);
#define SS2(V) hipLaunchKernelGGL((subsample2_kernel<T, V>), dim3(grid), dim3(256), 0, st, (const T*)src, (T*)dst, H, W, Ho, Wo, rows)
  switch (vec) {
    case 8: if constexpr (sizeof(T) == 2) { SS2(8); } break;
    case 4: SS2(4); break;
    case 2: SS2(2); break;
    default: SS2(1); break;
  }
#undef SS2
}

extern "C" int x3d_subsample2(const void* src, void* dst, long long planes, int H, int W, int dtype, void* stream) {
  X3D_REQUIRE(src && dst && planes > 0 && H > 0 && W > 0, "subsample2: bad args");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "subsample2: bad dtype");
  X3D_REQUIRE(ceil_div_ll(planes * ((H + 1) / 2) * ((W + 1) / 2), 256) < (1ll << 31), "subsample2: grid too large");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == X3D_F32) subsample2_launch<float>(src, dst, planes, H, W, st);
  else subsample2_launch<unsigned short>(src, dst, planes, H, W, st);      // (a copy: the 16-bit types move as bits)
  X3D_LAUNCH_CHECK("subsample2");
  return X3D_OK;
}
