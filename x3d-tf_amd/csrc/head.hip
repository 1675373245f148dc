// Classification head on pooled vectors (reference model.py:95-111,119-127; loss train.py:104) and
// the optimizer (train.py:89-92).  All fp32; sizes are [N <= a few hundred] x [432..2048] x [400].
#include "common.h"

// y[n][m] = act(sum_k xm[n][k]*w[m][k] + b[m]), xm = x * (mask ? mask*mask_scale : 1).
// One wave per DENSE_MT output features and DENSE_NT samples: lanes split K, coalesced on w[m][:] and x[n][:].
// The x rows are the traffic (every feature block re-reads them from L2: M/MT * N * K floats), so a wave keeps
// MT features: with one feature per wave fc2 (2048 -> 400, 64 samples) moved 210 MB through L1 in 105 us.
#define DENSE_NT 8
#define DENSE_MT 4
__global__ __launch_bounds__(64) void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                       float mask_scale, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* y, int act, int N, int K,
                                                       int M) {
  const int m0 = blockIdx.x * DENSE_MT, n0 = blockIdx.y * DENSE_NT, lane = threadIdx.x;
  float acc[DENSE_MT][DENSE_NT];
#pragma unroll
  for (int j = 0; j < DENSE_MT; j++)
#pragma unroll
    for (int i = 0; i < DENSE_NT; i++) acc[j][i] = 0.f;
  // rows past M / N are clamped (computed twice, never stored): every load of the loop is unconditional, so the
  // MT + NT (+ NT mask) loads of an iteration are in flight together instead of one branch at a time
  const float* wr[DENSE_MT];
  const float* xr[DENSE_NT];
  const float* mr[DENSE_NT];
#pragma unroll
  for (int j = 0; j < DENSE_MT; j++) wr[j] = w + (long long)min(m0 + j, M - 1) * K;
#pragma unroll
  for (int i = 0; i < DENSE_NT; i++) {
    xr[i] = x + (long long)min(n0 + i, N - 1) * K;
    mr[i] = mask ? mask + (long long)min(n0 + i, N - 1) * K : nullptr;
  }
  if (mask) {
#pragma unroll 2
    for (int k = lane; k < K; k += 64) {
      float wv[DENSE_MT], xv[DENSE_NT], mv[DENSE_NT];
#pragma unroll
      for (int j = 0; j < DENSE_MT; j++) wv[j] = wr[j][k];
#pragma unroll
      for (int i = 0; i < DENSE_NT; i++) { xv[i] = xr[i][k]; mv[i] = mr[i][k]; }
#pragma unroll
      for (int i = 0; i < DENSE_NT; i++) xv[i] *= mv[i] * mask_scale;
#pragma unroll
      for (int j = 0; j < DENSE_MT; j++)
#pragma unroll
        for (int i = 0; i < DENSE_NT; i++) acc[j][i] += wv[j] * xv[i];
    }
  } else {
#pragma unroll 2
    for (int k = lane; k < K; k += 64) {
      float wv[DENSE_MT], xv[DENSE_NT];
#pragma unroll
      for (int j = 0; j < DENSE_MT; j++) wv[j] = wr[j][k];
#pragma unroll
      for (int i = 0; i < DENSE_NT; i++) xv[i] = xr[i][k];
#pragma unroll
      for (int j = 0; j < DENSE_MT; j++)
#pragma unroll
        for (int i = 0; i < DENSE_NT; i++) acc[j][i] += wv[j] * xv[i];
    }
  }
#pragma unroll
  for (int j = 0; j < DENSE_MT; j++)
#pragma unroll
    for (int i = 0; i < DENSE_NT; i++) {
      const float s = wave_sum(acc[j][i]);
      const int n = n0 + i, m = m0 + j;
      if (lane == 0 && n < N && m < M) {
        float v = s + (b ? b[m] : 0.f);
        if (act == X3D_ACT_RELU) v = fmaxf(v, 0.f);
        y[(long long)n * M + m] = v;
      }
    }
}

// dx[n][k] = (sum_m dz[n][m]*w[m][k]) * (mask ? mask*scale : 1); dz = dy*[y>0] for ReLU.
// A workgroup owns 64 inputs k (the lanes) and DENSE_BN samples; its four waves split M (the long, latency-bound
// loop: 2048 features for fc1) and meet in LDS.  w[m][k] is read once per DENSE_BN samples.
#define DENSE_BN 4
__global__ __launch_bounds__(256) void dense_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           int act, const float* __restrict__ mask, float mask_scale,
                                                           const float* __restrict__ w, float* dx, int N, int K, int M) {
  extern __shared__ float dz[];  // [DENSE_BN][M], then the partial sums [4 waves][DENSE_BN][64]
  const int n0 = blockIdx.y * DENSE_BN;
  for (int i = threadIdx.x; i < DENSE_BN * M; i += 256) {
    const int nn = i / M, m = i - nn * M, n = n0 + nn;
    float d = 0.f;
    if (n < N) {
      d = dy[(long long)n * M + m];
      if (act == X3D_ACT_RELU && !(y[(long long)n * M + m] > 0.f)) d = 0.f;
    }
    dz[i] = d;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int kc = k < K ? k : K - 1;   // clamped: every lane loads, only k < K is stored
  float acc[DENSE_BN];
#pragma unroll
  for (int i = 0; i < DENSE_BN; i++) acc[i] = 0.f;
  // batches of 16 loads issued together (written out: left to the unroller, every load was followed by vmcnt(0))
  constexpr int DXB = 16;
  int m = wid;
  for (; m + 4 * (DXB - 1) < M; m += 4 * DXB) {
    float wv[DXB];
#pragma unroll
    for (int j = 0; j < DXB; j++) wv[j] = w[(long long)(m + 4 * j) * K + kc];
#pragma unroll
    for (int j = 0; j < DXB; j++)
#pragma unroll
      for (int i = 0; i < DENSE_BN; i++) acc[i] += dz[i * M + m + 4 * j] * wv[j];   // dz: LDS broadcast reads
  }
  for (; m < M; m += 4) {
    const float wv = w[(long long)m * K + kc];
#pragma unroll
    for (int i = 0; i < DENSE_BN; i++) acc[i] += dz[i * M + m] * wv;
  }
  __syncthreads();   // everyone is done with dz
#pragma unroll
  for (int i = 0; i < DENSE_BN; i++) dz[(wid * DENSE_BN + i) * 64 + lane] = acc[i];
  __syncthreads();
  if (wid == 0 && k < K) {
#pragma unroll
    for (int i = 0; i < DENSE_BN; i++) {
      const int n = n0 + i;
      if (n < N) {
        float v = dz[i * 64 + lane] + dz[(DENSE_BN + i) * 64 + lane] + dz[(2 * DENSE_BN + i) * 64 + lane] +
                  dz[(3 * DENSE_BN + i) * 64 + lane];
        if (mask) v *= mask[(long long)n * K + k] * mask_scale;
        dx[(long long)n * K + k] = v;
      }
    }
  }
}

// dw[m][k] += sum_n dz[n][m]*xm[n][k] ; db[m] += sum_n dz[n][m].
// A workgroup owns 256 inputs k and DENSE_BM features: xm[n][k] is read once per DENSE_BM features.
#define DENSE_BM 8
__global__ __launch_bounds__(256) void dense_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           int act, const float* __restrict__ x,
                                                           const float* __restrict__ mask, float mask_scale, float* dw,
                                                           float* db, int N, int K, int M) {
  extern __shared__ float dzs[];   // [N rounded up to 8][DENSE_BM]: dz of this workgroup's features, ReLU mask applied, 0 past N
  const int m0 = blockIdx.y * DENSE_BM;
  const int Np = (N + 7) & ~7;
  for (int i = threadIdx.x; i < Np * DENSE_BM; i += 256) {
    const int n = i / DENSE_BM, m = m0 + (i - n * DENSE_BM);
    float d = 0.f;
    if (m < M && n < N) {
      d = dy[(long long)n * M + m];
      if (act == X3D_ACT_RELU && !(y[(long long)n * M + m] > 0.f)) d = 0.f;
    }
    dzs[i] = d;
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int kc = k < K ? k : K - 1;   // clamped: every lane loads (unconditional, 8 samples in flight), only k < K is stored
  float acc[DENSE_BM], accb[DENSE_BM];
#pragma unroll
  for (int j = 0; j < DENSE_BM; j++) { acc[j] = 0.f; accb[j] = 0.f; }
  // samples in batches of 8 with the loads written out first (left to the unroller, the masked variant waited for
  // every load separately); rows past N are clamped and weighted by dz = 0
  constexpr int DWB = 8;
  for (int nb = 0; nb < N; nb += DWB) {
    float xv[DWB], mv[DWB];
#pragma unroll
    for (int u = 0; u < DWB; u++) xv[u] = x[(long long)min(nb + u, N - 1) * K + kc];
    if (mask) {
#pragma unroll
      for (int u = 0; u < DWB; u++) mv[u] = mask[(long long)min(nb + u, N - 1) * K + kc];
#pragma unroll
      for (int u = 0; u < DWB; u++) xv[u] *= mv[u] * mask_scale;
    }
#pragma unroll
    for (int u = 0; u < DWB; u++) {
      const f32x4 d0 = *(const f32x4*)&dzs[(nb + u) * DENSE_BM], d1 = *(const f32x4*)&dzs[(nb + u) * DENSE_BM + 4];   // LDS broadcast
#pragma unroll
      for (int j = 0; j < 4; j++) {
        accb[j] += d0[j]; acc[j] += d0[j] * xv[u];
        accb[4 + j] += d1[j]; acc[4 + j] += d1[j] * xv[u];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < DENSE_BM; j++) {
    const int m = m0 + j;
    if (m < M) {
      if (k < K) dw[(long long)m * K + k] += acc[j];
      if (db && blockIdx.x == 0 && threadIdx.x == 0) db[m] += accb[j];
    }
  }
}

extern "C" int x3d_dense_fwd(const float* x, const float* mask, float mask_scale, const float* w, const float* b,
                             float* y, int act, int N, int K, int M, void* stream) {
  X3D_REQUIRE(x && w && y && N > 0 && K > 0 && M > 0, "dense_fwd: bad args");
  X3D_REQUIRE(act == X3D_ACT_NONE || act == X3D_ACT_RELU, "dense_fwd: act must be none/relu");
  hipLaunchKernelGGL(dense_fwd_kernel, dim3(ceil_div(M, DENSE_MT), ceil_div(N, DENSE_NT)), dim3(64), 0, (hipStream_t)stream,
                     x, mask, mask_scale, w, b, y, act, N, K, M);
  X3D_LAUNCH_CHECK("dense_fwd");
  return X3D_OK;
}

extern "C" int x3d_dense_bwd(const float* dy, const float* y, int act, const float* x, const float* mask,
                             float mask_scale, const float* w, float* dx, float* dw, float* db, int N, int K, int M,
                             void* stream) {
  X3D_REQUIRE(dy && x && w && dw && N > 0 && K > 0 && M > 0, "dense_bwd: bad args");
  X3D_REQUIRE(act == X3D_ACT_NONE || (act == X3D_ACT_RELU && y), "dense_bwd: relu needs y");
  const size_t dz_bytes = (size_t)DENSE_BN * (M > 256 ? M : 256) * sizeof(float);   // >= the [4][DENSE_BN][64] partial sums
  X3D_REQUIRE(dz_bytes <= 64 * 1024, "dense_bwd: M too large");
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    hipLaunchKernelGGL(dense_bwd_dx_kernel, dim3(ceil_div(K, 64), ceil_div(N, DENSE_BN)), dim3(256), dz_bytes, st, dy, y, act,
                       mask, mask_scale, w, dx, N, K, M);
    X3D_LAUNCH_CHECK("dense_bwd_dx");
  }
  const size_t dzs_bytes = (size_t)((N + 7) & ~7) * DENSE_BM * sizeof(float);
  X3D_REQUIRE(dzs_bytes <= 64 * 1024, "dense_bwd: N too large");
  hipLaunchKernelGGL(dense_bwd_dw_kernel, dim3(ceil_div(K, 256), ceil_div(M, DENSE_BM)), dim3(256), dzs_bytes, st, dy, y, act, x, mask, mask_scale, dw, db, N, K, M);
  X3D_LAUNCH_CHECK("dense_bwd_dw");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// softmax + Keras sparse categorical cross-entropy on probabilities.  One block per sample.
//   p = softmax(z);  q = clip(p, 1e-7, 1-1e-7);  L = -log q_y + log sum_j q_j          [TF-3p]
//   dL/dp_j = [1e-7 <= p_j <= 1-1e-7] * (-[j==y]/q_y + 1/sum q);   dz = p * (dL/dp - sum_k p_k dL/dp_k)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits,
                                                           const int* __restrict__ labels, float* probs,
                                                           float* loss_rows, float* dlogits, float grad_scale, int M) {
  __shared__ float scratch[8];
  __shared__ float bc[2];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (long long)n * M;
  float mx = -INFINITY;
  for (int j = tid; j < M; j += 256) mx = fmaxf(mx, z[j]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) scratch[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
  __syncthreads();
  float red[1] = {0.f};
  for (int j = tid; j < M; j += 256) red[0] += expf(z[j] - mx);
  block_sum<1>(red, scratch);
  if (tid == 0) bc[0] = red[0];
  __syncthreads();
  const float inv = 1.f / bc[0];
  __syncthreads();
  const int label = labels ? labels[n] : -1;
  float r2[2] = {0.f, 0.f};  // sum q, sum_k p_k*[in range]  (for the dz formula)
  for (int j = tid; j < M; j += 256) {
    const float p = expf(z[j] - mx) * inv;
    probs[(long long)n * M + j] = p;
    const float q = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    r2[0] += q;
    r2[1] += (p >= 1e-7f && p <= 1.f - 1e-7f) ? p : 0.f;
  }
  if (!labels) return;
  block_sum<2>(r2, scratch);
  if (tid == 0) { bc[0] = r2[0]; bc[1] = r2[1]; }
  __syncthreads();
  const float sumq = bc[0], sum_p_in = bc[1];
  if (label < 0 || label >= M) {
    // out-of-range label (TF raises InvalidArgument): never index with it -- the loss row is NaN (so the step's loss is
    // visibly non-finite) and the sample contributes no gradient
    if (tid == 0 && loss_rows) loss_rows[n] = __builtin_nanf("");
    if (dlogits)
      for (int j = tid; j < M; j += 256) dlogits[(long long)n * M + j] = 0.f;
    return;
  }
  const float py = expf(z[label] - mx) * inv;
  const float qy = fminf(fmaxf(py, 1e-7f), 1.f - 1e-7f);
  const bool y_in = (py >= 1e-7f && py <= 1.f - 1e-7f);
  if (tid == 0 && loss_rows) loss_rows[n] = -logf(qy) + logf(sumq);
  if (dlogits) {
    // inner = sum_k p_k * dL/dp_k = sum_p_in / sumq - [y in range] * p_y / q_y
    const float inner = sum_p_in / sumq - (y_in ? py / qy : 0.f);
    for (int j = tid; j < M; j += 256) {
      const float p = expf(z[j] - mx) * inv;
      const bool in = (p >= 1e-7f && p <= 1.f - 1e-7f);
      float dldp = in ? 1.f / sumq : 0.f;
      if (j == label && in) dldp -= 1.f / qy;
      dlogits[(long long)n * M + j] = grad_scale * p * (dldp - inner);
    }
  }
}

extern "C" int x3d_softmax_xent(const float* logits, const int* labels, float* probs, float* loss_rows,
                                float* dlogits, float grad_scale, int N, int M, void* stream) {
  X3D_REQUIRE(logits && probs && N > 0 && M > 0, "softmax_xent: bad args");
  X3D_REQUIRE(labels || (!loss_rows && !dlogits), "softmax_xent: loss/grad need labels");
  hipLaunchKernelGGL(softmax_xent_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, logits, labels, probs,
                     loss_rows, dlogits, grad_scale, M);
  X3D_LAUNCH_CHECK("softmax_xent");
  return X3D_OK;
}

__global__ void view_mean_kernel(const float* __restrict__ probs, float* out, int views, int M) {
  const int v = blockIdx.y;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float acc = 0.f;
  for (int i = 0; i < views; i++) acc += probs[((long long)v * views + i) * M + m];
  out[(long long)v * M + m] = acc / (float)views;
}

extern "C" int x3d_view_mean(const float* probs, float* out, int videos, int views, int M, void* stream) {
  X3D_REQUIRE(probs && out && videos > 0 && views > 0 && M > 0, "view_mean: bad args");
  hipLaunchKernelGGL(view_mean_kernel, dim3(ceil_div(M, 128), videos), dim3(128), 0, (hipStream_t)stream, probs, out,
                     views, M);
  X3D_LAUNCH_CHECK("view_mean");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// SGD + Nesterov momentum + L2 (flat arrays)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* __restrict__ w, float* __restrict__ v,
                                                           const float* __restrict__ g,
                                                           const unsigned char* __restrict__ l2, float lr, float mom,
                                                           float wd, float gscale, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gi = g[i] * gscale;
  const float wi = w[i];
  if (l2 && l2[i]) gi += 2.f * wd * wi;
  const float vi = mom * v[i] - lr * gi;
  v[i] = vi;
  w[i] = wi + mom * vi - lr * gi;
}

extern "C" int x3d_sgd_nesterov(float* w, float* v, const float* g, const unsigned char* l2_mask, float lr,
                                float momentum, float weight_decay, float grad_scale, long long n, void* stream) {
  X3D_REQUIRE(w && v && g && n > 0, "sgd_nesterov: bad args");
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     v, g, l2_mask, lr, momentum, weight_decay, grad_scale, n);
  X3D_LAUNCH_CHECK("sgd_nesterov");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// Adam (tf.optimizers.Adam(learning_rate), the reference's other optimizer branch, train.py:93-95; Keras defaults
// beta_1 = 0.9, beta_2 = 0.999, epsilon = 1e-7, no amsgrad):  g' = g*grad_scale + 2*wd*w (where l2_mask)
//   m = b1*m + (1-b1)*g' ; v = b2*v + (1-b2)*g'^2 ; w -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)        [TF-3p]
// One launch over the flat parameter buffer, like x3d_sgd_nesterov.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, const unsigned char* __restrict__ l2,
                                                   float lr_t, float b1, float b2, float eps, float wd, float gscale,
                                                   long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float gi = g[i] * gscale;
  const float wi = w[i];
  if (l2 && l2[i]) gi += 2.f * wd * wi;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
}

extern "C" int x3d_adam(float* w, float* m, float* v, const float* g, const unsigned char* l2_mask, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float grad_scale, long long step, long long n,
                        void* stream) {
  X3D_REQUIRE(w && m && v && g && n > 0 && step >= 1, "adam: bad args (step counts from 1)");
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)ceil_div_ll(n, 256)), dim3(256), 0, (hipStream_t)stream, w, m, v, g,
                     l2_mask, (float)lr_t, beta1, beta2, eps, weight_decay, grad_scale, n);
  X3D_LAUNCH_CHECK("adam");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// LossScaleOptimizer support (train.py:99-100, Keras mixed_float16): are all gradients finite?  *flag (device int,
// set to 1 by the caller) is cleared when any of the n values is inf / nan.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void all_finite_kernel(const float* __restrict__ g, long long n, int* flag) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const unsigned u = __float_as_uint(g[i]);
    bad |= (u & 0x7f800000u) == 0x7f800000u;     // exponent all ones: inf or nan
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

extern "C" int x3d_all_finite(const float* g, long long n, int* flag, void* stream) {
  X3D_REQUIRE(g && flag && n > 0, "all_finite: bad args");
  long long blocks = ceil_div_ll(n, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(all_finite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, flag);
  X3D_LAUNCH_CHECK("all_finite");
  return X3D_OK;
}

__global__ __launch_bounds__(256) void l2_sumsq_kernel(const float* __restrict__ w,
                                                       const unsigned char* __restrict__ l2, double* out, long long n) {
  __shared__ float scratch[4];
  float red[1] = {0.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    if (!l2 || l2[i]) red[0] += w[i] * w[i];
  block_sum<1>(red, scratch);
  if (threadIdx.x == 0) atomic_add_d(out, (double)red[0]);
}

extern "C" int x3d_l2_sumsq(const float* w, const unsigned char* l2_mask, double* out, long long n, void* stream) {
  X3D_REQUIRE(w && out && n > 0, "l2_sumsq: bad args");
  long long blocks = ceil_div_ll(n, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(l2_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, l2_mask, out, n);
  X3D_LAUNCH_CHECK("l2_sumsq");
  return X3D_OK;
}
