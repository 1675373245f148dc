// Pointwise (1x1x1) convolutions as GEMMs over points on the MI355X matrix cores.
//
// NCTHW makes every sample a [C][P] row-major matrix with the P = T*H*W points contiguous, so
//   forward  Y[n] = W   * f(X[n])        M = Cout, K = Cin , N = points
//   dgrad    dX[n] = W^T * dYraw[n]      M = Cin , K = Cout, N = points
//   wgrad    dW   += dYraw[n] * f(X[n])^T M = Cout, N = Cin , K = points
// are im2col-free.  This file is the exact-fp32 path: v_mfma_f32_32x32x2_f32 takes ONE f32 per lane
// per operand with lane = (k = lane>>5, row/col = lane&31), so the [k][32 points] operand is read
// from an LDS tile whose rows are points-contiguous without any transpose, and the 32x32 result has
// the point index on the lane (stores/loads in the epilogue touch 32 consecutive points per
// half-wave).  Tiles are staged global -> registers -> (prologue transform) -> LDS once per element,
// so the folded BN / SE gate / activation / BN-backward arithmetic runs once per element, not once
// per MFMA operand read.
#pragma once
#include "common.h"

// PRO_TAIL (forward, 16-bit storage): the conv input is the residual tail of the block below, built on load --
//   v = relu(s_c * x + t_c + (s_r * x2 + t_r | x2)),  x = that block's raw `c` output, x2 = its shortcut (raw shortcut-conv
//   output or its input) -- and stored to `ystore` as the block's output y: no separate x3d_tail_fwd pass.  Same two-tensor
//   staging as PRO_BNBWD (v = A*x + B*x2 + C per row) + ReLU + the side store.
// PRO_AFFST (forward, 16-bit storage): PRO_AFFINE whose activated input v = act(s * x + t) is also stored to `ystore` -- the
//   stem's BatchNorm + ReLU (x = the raw temporal-conv output, reference model.py:202-210) built by its first reader, the
//   `a` conv of the first residual block, instead of by a pass of its own.
enum { PRO_NONE = 0, PRO_AFFINE = 1, PRO_BNBWD = 2, PRO_TAIL = 3, PRO_AFFST = 4 };

enum { EPI_STATS = 100,     // forward epilogue (training): store raw + per-channel statistics
       EPI_BNADD = 101 };   // forward epilogue (inference): y = act(s_o*acc + t_o [+ s_r*add + t_r]) -- folded BN, residual Add + ReLU

struct PwGemmArgs {
  // streamed operand  [N][K][Pin]
  const void* x;
  const void* x2;      // PRO_BNBWD: raw conv output
  const float* coef;   // PRO_AFFINE: [K][2] ; PRO_BNBWD: [K][4] ; PRO_TAIL: [K][2] (scale, shift of x)
  const float* coef2;  // PRO_TAIL: [K][2] (scale, shift of x2) or null (x2 taken as it is)
  void* ystore;        // PRO_TAIL / PRO_AFFST: [N][K][Pin] the activated input, written by the first row-block slice
  const float* gate;   // PRO_AFFINE: [N][K] or null
  int act;
  // weights, element (k, m) at w[k*wsk + m*wsm]
  const float* w;
  int wsk, wsm;
  int N, K, M;
  long long P, Pin;    // points per sample of the output side (y, add, braw) / of the streamed operand x, x2
  int stride, H, W, Ho, Wo;  // strided gather (stride > 1): source H,W ; sampled Ho,Wo
  int KC, nchunks, tiles_per_block;
  int wvec;            // bf16 kernel: fp32 vector width for the weight-panel staging
  BnBwdFold fold;      // PRO_BNBWD, pw_gemm_wst.h only: sums != NULL -> `coef` is derived from the BatchNorm-backward sums (x3d_hip.h coef_fold)
  int hot;             // experiments build only (X3D_PW_WST_HOT=1, pw_gemm_wst.h): every tile load re-reads the first tile
  const void* wp;      // bf16 kernel: packed panel (x3d_pw_pack_weights) [wp_rows][KC + 8] or null
  int wp_rows;
  // epilogue
  void* y;
  double* stats;       // EPI_STATS: [M][2]
  const void* add;
  const void* braw;
  const float* b_ss;
  const float* egate;
  double* nc_sums;
  int eH, eW;          // EPI_ADD_STRIDED: geometry of dx (H, W); add is at ceil(H/2) x ceil(W/2)
  // EPI_BNADD: per-output-row (scale, shift) [M][2]; `add` [N][M][P] or null with its own optional (scale, shift) [M][2];
  // eact = X3D_ACT_NONE / X3D_ACT_RELU
  const float* e_ss;
  const float* e_ass;
  int eact;
};

// EPI_BNADD coefficients of output row m: v = c0*acc + c1 + c2*add
__device__ __forceinline__ void bnadd_coef(const PwGemmArgs& a, int m, bool ok, float& c0, float& c1, float& c2) {
  c0 = ok ? a.e_ss[m * 2] : 0.f;
  c1 = ok ? a.e_ss[m * 2 + 1] : 0.f;
  c2 = 1.0f;
  if (ok && a.e_ass) { c2 = a.e_ass[m * 2]; c1 += a.e_ass[m * 2 + 1]; }
}

// the same in two halves -- raw loads, then the arithmetic -- so that a thread can put ALL its staging loads of a tile in
// flight before it touches the first value (stage_x below)
template <typename T, int VEC, int PRO, bool STRIDED>
__device__ __forceinline__ void pw_load_raw(const PwGemmArgs& a, int n, int gk, long long p, float (&v)[VEC], float (&y2)[VEC]) {
  const T* x = (const T*)a.x;
  if constexpr (STRIDED) {
    static_assert(VEC == 1, "strided gather is scalar");
    long long hw = (long long)a.Ho * a.Wo;
    long long t = p / hw;
    int rem = (int)(p - t * hw);
    int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    long long src = (t * a.H + (long long)ho * a.stride) * a.W + (long long)wo * a.stride;
    v[0] = to_f<T>(x[((long long)n * a.K + gk) * a.Pin + src]);
  } else {
    VecIO<T, VEC>::load(x + ((long long)n * a.K + gk) * a.Pin + p, v);
  }
  if constexpr (PRO == PRO_BNBWD) VecIO<T, VEC>::load((const T*)a.x2 + ((long long)n * a.K + gk) * a.Pin + p, y2);
}
template <int VEC, int PRO>
__device__ __forceinline__ void pw_prologue(const PwGemmArgs& a, const float* ck, float (&v)[VEC], const float (&y2)[VEC]) {
  if constexpr (PRO == PRO_AFFINE) {
    const float s = ck[0], t = ck[1], g = ck[2];
#pragma unroll
    for (int e = 0; e < VEC; e++) v[e] = (s * v[e] + t) * g;
    act_vec<VEC>(v, a.act);
  } else if constexpr (PRO == PRO_BNBWD) {
    const float A = ck[0], B = ck[1], C = ck[2];
#pragma unroll
    for (int e = 0; e < VEC; e++) v[e] = A * v[e] + B * y2[e] + C;
  }
}

// ck: this row's prologue coefficients from the workgroup's LDS table (PRO_AFFINE: {s*g, t*g}; PRO_BNBWD: {A, B, C})
template <typename T, int VEC, int PRO, bool STRIDED>
__device__ __forceinline__ void pw_load_vec(const PwGemmArgs& a, int n, int gk, long long p,
                                            const float* ck, float (&v)[VEC]) {
  const T* x = (const T*)a.x;
  if constexpr (STRIDED) {
    static_assert(VEC == 1, "strided gather is scalar");
    long long hw = (long long)a.Ho * a.Wo;
    long long t = p / hw;
    int rem = (int)(p - t * hw);
    int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    long long src = (t * a.H + (long long)ho * a.stride) * a.W + (long long)wo * a.stride;
    v[0] = to_f<T>(x[((long long)n * a.K + gk) * a.Pin + src]);
  } else {
    VecIO<T, VEC>::load(x + ((long long)n * a.K + gk) * a.Pin + p, v);
  }
  if constexpr (PRO == PRO_AFFINE) {
    const float s = ck[0], t = ck[1], g = ck[2];
#pragma unroll
    for (int e = 0; e < VEC; e++) v[e] = (s * v[e] + t) * g;
    act_vec<VEC>(v, a.act);
  } else if constexpr (PRO == PRO_BNBWD) {
    float y2[VEC];
    VecIO<T, VEC>::load((const T*)a.x2 + ((long long)n * a.K + gk) * a.Pin + p, y2);
    const float A = ck[0], B = ck[1], C = ck[2];
#pragma unroll
    for (int e = 0; e < VEC; e++) v[e] = A * v[e] + B * y2[e] + C;
  }
}

template <typename T, int VEC, int MT, int NT, int PRO, int EPI, bool STRIDED>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const PwGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BM = MT * 32, BN = NT * 32;
  constexpr int TPW = (MT * NT + 3) / 4;
  constexpr bool HAS_SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  constexpr int BMP = BM + 1;      // odd W pitch: staging along k (stride BMP) is bank-conflict-free
  float* Xs = smem;                // [KC][BN]   (16-B aligned for the vector stores)
  float* Ws = smem + a.KC * BN;    // [KC][BMP]
  // per-row coefficients as LDS tables, filled once per workgroup (chunk): read from global inside the staging loop
  // and the epilogue they were 3 dependent L2 round trips per vector / per output element
  float* Pk = Ws + a.KC * BMP;     // [KC][4]  prologue rows (PRO_AFFINE / PRO_BNBWD)
  float* Em = Pk + a.KC * 4;       // [BM][4]  epilogue rows (SWISH_BWD: {s_b, t_b, gate})

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * BM;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int chunks_per_n = (tiles_per_n + a.tiles_per_block - 1) / a.tiles_per_block;
  const int n = blockIdx.x / chunks_per_n;
  const int chunk = blockIdx.x - n * chunks_per_n;
  const int tile_begin = chunk * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, tiles_per_n);

  f32x16 acc[TPW];
  float st1[HAS_SUMS ? TPW : 1][16], st2[HAS_SUMS ? TPW : 1][16];
  if constexpr (HAS_SUMS) {
#pragma unroll
    for (int s = 0; s < TPW; s++)
#pragma unroll
      for (int j = 0; j < 16; j++) { st1[s][j] = 0.f; st2[s][j] = 0.f; }
  }

  auto stage_w = [&](int k0) {
    // walk the weight matrix along its contiguous axis so the global reads coalesce.  EIGHT elements per thread and round,
    // all eight loads issued before the first LDS write: the rolled loop (run-time trip count, an integer division per
    // element) waited for every load in turn -- 24-48 exposed L2 latencies per panel, which was most of the run time of the
    // small layers (216 -> 96 on 8000 points: 69 us)
    constexpr int UW = 8;
    const int total = a.KC * BM;
    const int kshift = 32 - __builtin_clz((unsigned)(a.KC - 1) | 1u);   // KC rounded up to a power of two: k = i & mask
    const int kmask = (1 << kshift) - 1;
    const int span = (a.wsk == 1) ? (BM << kshift) : total;
    for (int base = 0; base < span; base += 256 * UW) {
      float wv[UW];
      int dst[UW];
#pragma unroll
      for (int u = 0; u < UW; u++) {
        const int i = base + u * 256 + tid;
        int k, m;
        if (a.wsk == 1) { m = i >> kshift; k = i & kmask; }
        else { k = i / BM; m = i - k * BM; }
        const bool in = i < span && k < a.KC && m < BM;
        const int gk = k0 + k, gm = m0 + m;
        dst[u] = in ? k * BMP + m : -1;
        wv[u] = (in && gk < a.K && gm < a.M) ? a.w[(long long)gk * a.wsk + (long long)gm * a.wsm] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UW; u++) if (dst[u] >= 0) Ws[dst[u]] = wv[u];
    }
  };
  auto stage_x = [&](int k0, long long p0) {
    constexpr int VPR = BN / VEC;  // vectors per row
    // four vectors per thread and round, their loads issued together (the rolled loop waited for each in turn)
    constexpr int UX = 4;
    const int nvec = a.KC * VPR;
    for (int base = 0; base < nvec; base += 256 * UX) {
      float xr[UX][VEC], yr[PRO == PRO_BNBWD ? UX : 1][VEC];
#pragma unroll
      for (int u = 0; u < UX; u++) {
        const int v = base + u * 256 + tid;
        const int k = v / VPR, pv = v - k * VPR;
        const long long p = p0 + (long long)pv * VEC;
#pragma unroll
        for (int e = 0; e < VEC; e++) { xr[u][e] = 0.f; if constexpr (PRO == PRO_BNBWD) yr[u][e] = 0.f; }
        if (v < nvec && k0 + k < a.K && p < a.P) pw_load_raw<T, VEC, PRO, STRIDED>(a, n, k0 + k, p, xr[u], yr[PRO == PRO_BNBWD ? u : 0]);
      }
#pragma unroll
      for (int u = 0; u < UX; u++) {
        const int v = base + u * 256 + tid;
        if (v >= nvec) continue;
        const int k = v / VPR, pv = v - k * VPR;
        const long long p = p0 + (long long)pv * VEC;
        float val[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) val[e] = xr[u][e];
        if (k0 + k < a.K && p < a.P) {
          pw_prologue<VEC, PRO>(a, Pk + k * 4, val, yr[PRO == PRO_BNBWD ? u : 0]);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; e++) val[e] = 0.f;   // the prologue constants must not leak into the padding
        }
        VecIO<float, VEC>::store(&Xs[k * BN + pv * VEC], val);
      }
    }
    return;
    for (int v = tid; v < a.KC * VPR; v += 256) {
      int k = v / VPR, pv = v - k * VPR;
      int gk = k0 + k;
      long long p = p0 + (long long)pv * VEC;
      float val[VEC];
      if (gk < a.K && p < a.P) {
        pw_load_vec<T, VEC, PRO, STRIDED>(a, n, gk, p, Pk + k * 4, val);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) val[e] = 0.f;
      }
      VecIO<float, VEC>::store(&Xs[k * BN + pv * VEC], val);
    }
  };

  auto fill_pk = [&](int k0) {
    if constexpr (PRO != PRO_NONE) {
      for (int k = tid; k < a.KC; k += 256) {
        const int gk = k0 + k;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (gk < a.K) {
          if constexpr (PRO == PRO_AFFINE) {
            c0 = a.coef[gk * 2]; c1 = a.coef[gk * 2 + 1];
            c2 = a.gate ? a.gate[(long long)n * a.K + gk] : 1.0f;
          } else {
            c0 = a.coef[gk * 4]; c1 = a.coef[gk * 4 + 1]; c2 = a.coef[gk * 4 + 2];
          }
        }
        Pk[k * 4] = c0; Pk[k * 4 + 1] = c1; Pk[k * 4 + 2] = c2;
      }
    }
  };
  if constexpr (EPI == X3D_EPI_SWISH_BWD) {
    for (int m = tid; m < BM; m += 256) {
      const int gm = m0 + m;
      const bool ok = gm < a.M;
      Em[m * 4] = ok ? a.b_ss[gm * 2] : 0.f;
      Em[m * 4 + 1] = ok ? a.b_ss[gm * 2 + 1] : 0.f;
      Em[m * 4 + 2] = (ok && a.egate) ? a.egate[(long long)n * a.M + gm] : 1.0f;
    }
  }
  if constexpr (EPI == EPI_BNADD) {
    for (int m = tid; m < BM; m += 256) {
      float c0, c1, c2;
      bnadd_coef(a, m0 + m, m0 + m < a.M, c0, c1, c2);
      Em[m * 4] = c0; Em[m * 4 + 1] = c1; Em[m * 4 + 2] = c2;
    }
  }
  if (a.nchunks == 1) { stage_w(0); fill_pk(0); __syncthreads(); }

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const long long p0 = (long long)tile * BN;
#pragma unroll
    for (int s = 0; s < TPW; s++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

    for (int kc = 0; kc < a.nchunks; ++kc) {
      const int k0 = kc * a.KC;
      __syncthreads();  // every wave is done reading the previous tiles
      if (a.nchunks > 1) {
        stage_w(k0);
        if constexpr (PRO != PRO_NONE) { fill_pk(k0); __syncthreads(); }
      }
      stage_x(k0, p0);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        const int id = wid + 4 * s;
        if (id < MT * NT) {
          const int mt = id / NT, nt = id - mt * NT;
          const float* wp = Ws + half * BMP + mt * 32 + r;
          const float* xp = Xs + half * BN + nt * 32 + r;
          for (int kk = 0; kk < a.KC; kk += 2) {
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[kk * BMP], xp[kk * BN], acc[s], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue: D[row][col]: col = lane&31 (point), row = (j&3) + 8*(j>>2) + 4*(lane>>5)
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + 4 * s;
      if (id < MT * NT) {
        const int mt = id / NT, nt = id - mt * NT;
        const long long p = p0 + nt * 32 + r;
        // the epilogue operand (residual / raw depthwise output) of all 16 rows first: inside the loop below every load
        // sat behind the previous row's store (the compiler must assume they alias) and exposed its latency 16 times
        constexpr bool EPL = (EPI == X3D_EPI_ADD) || (EPI == X3D_EPI_SWISH_BWD) || (EPI == EPI_BNADD);
        float eop[EPL ? 16 : 1];
        if constexpr (EPL) {
          const T* esrc = (const T*)(EPI == X3D_EPI_SWISH_BWD ? a.braw : a.add);
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const int m = m0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
            eop[j] = ((m < a.M) && (p < a.P) && esrc) ? to_f<T>(esrc[((long long)n * a.M + m) * a.P + p]) : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int m = m0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
          const bool ok = (m < a.M) && (p < a.P);
          float val = acc[s][j];
          const long long o = ((long long)n * a.M + m) * a.P + p;
          if constexpr (EPI == EPI_STATS) {
            if (ok) {
              ((T*)a.y)[o] = from_f<T>(val);
              float vr = round_to<T>(val);
              st1[s][j] += vr;
              st2[s][j] += vr * vr;
            }
          } else if constexpr (EPI == X3D_EPI_STORE) {
            if (ok) ((T*)a.y)[o] = from_f<T>(val);
          } else if constexpr (EPI == EPI_BNADD) {
            if (ok) {
              const float* em = Em + (m - m0) * 4;
              float v = em[0] * val + em[1] + em[2] * eop[j];
              if (a.eact == X3D_ACT_RELU) v = fmaxf(v, 0.f);
              ((T*)a.y)[o] = from_f<T>(v);
            }
          } else if constexpr (EPI == X3D_EPI_ADD) {
            if (ok) ((T*)a.y)[o] = from_f<T>(val + eop[j]);
          } else if constexpr (EPI == X3D_EPI_ADD_STRIDED) {
            if (ok) {
              const long long hw = (long long)a.eH * a.eW;
              const long long t = p / hw;
              const int rem = (int)(p - t * hw);
              const int h = rem / a.eW, w = rem - h * a.eW;
              if (((h | w) & 1) == 0) {
                const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
                const long long T_ = a.P / hw;
                const long long oa =
                    ((((long long)n * a.M + m) * T_ + t) * Hh + (h >> 1)) * Wh + (w >> 1);
                val += to_f<T>(((const T*)a.add)[oa]);
              }
              ((T*)a.y)[o] = from_f<T>(val);
            }
          } else if constexpr (EPI == X3D_EPI_SWISH_BWD) {
            if (ok) {
              const float b = eop[j];
              const float* em = Em + (m - m0) * 4;
              const float u = em[0] * b + em[1];
              const float g = em[2];
              const float dv = val * swish_grad_(u * g);
              ((T*)a.y)[o] = from_f<T>(dv);
              const float dvr = round_to<T>(dv);
              st1[s][j] += dvr;
              st2[s][j] += dvr * b;
            }
          }
        }
      }
    }
  }

  if constexpr (HAS_SUMS) {
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + 4 * s;
      if (id < MT * NT) {
        const int mt = id / NT;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const float s1 = half_wave_sum_hi(st1[s][j]);
          const float s2 = half_wave_sum_hi(st2[s][j]);
          const int m = m0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
          if (r == 16 && m < a.M) {
            if constexpr (EPI == EPI_STATS) {
              if (a.stats) {
                double* sp = stats_replica(a.stats, a.M, blockIdx.x);
                atomic_add_d(&sp[m * 2], (double)s1);
                atomic_add_d(&sp[m * 2 + 1], (double)s2);
              }
            } else {
              double* d = a.nc_sums + ((long long)n * a.M + m) * 2;
              atomic_add_d(d, (double)s1);
              atomic_add_d(d + 1, (double)s2);
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int MT, int NT, int PRO, int EPI, bool STRIDED>
static int pw_launch_cfg(PwGemmArgs& a, hipStream_t st) {
  constexpr int BM = MT * 32, BN = NT * 32;
  const int Kpad = (a.K + 1) & ~1;
  const size_t resident = (size_t)Kpad * (BM + 1 + BN + 4) * sizeof(float) + (size_t)BM * 16;
  if (resident <= 80 * 1024) {
    a.KC = Kpad;
    a.nchunks = 1;
  } else {
    a.KC = 64;
    a.nchunks = ceil_div(a.K, 64);
  }
  const size_t lds = (size_t)a.KC * (BM + 1 + BN + 4) * sizeof(float) + (size_t)BM * 16;
  const int gy = ceil_div(a.M, BM);
  const long long tiles_per_n = ceil_div_ll(a.P, BN);
  const long long total = tiles_per_n * a.N * gy;
  X3D_DESCRIBE("pw_gemm_kernel<float, %d, %d, %d, %d, %d, %d>", VEC, MT, NT, PRO, EPI, (int)STRIDED);
  auto kern = pw_gemm_kernel<T, VEC, MT, NT, PRO, EPI, STRIDED>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_set = true;
    }
  }
  // one round of workgroups: every workgroup stages the whole weight panel (tens of KB) before its first tile, so a
  // layer with few points per sample (1344 one-tile workgroups on 512 slots) spent more on weights than on activations
  static size_t occ_lds[4];
  static int occ_slots[4], occ_n = 0;
  int slots = 0;
  for (int i = 0; i < occ_n; i++) if (occ_lds[i] == lds) slots = occ_slots[i];
  if (slots == 0) {
    int nb = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    slots = nb * cus;
    if (occ_n < 4) { occ_lds[occ_n] = lds; occ_slots[occ_n] = slots; occ_n++; }
  }
  long long tpb = ceil_div_ll(total, slots);
  if (tpb < 1) tpb = 1;
  if (tpb > 16) tpb = 16;
  if (tpb > tiles_per_n) tpb = tiles_per_n;
  a.tiles_per_block = (int)tpb;
  const long long gx = ceil_div_ll(tiles_per_n, tpb) * a.N;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_gemm");
  return X3D_OK;
}

template <typename T, int VEC, int PRO, int EPI, bool STRIDED>
static int pw_launch_tile(PwGemmArgs& a, hipStream_t st) {
  if (a.M <= 32) return pw_launch_cfg<T, VEC, 1, 4, PRO, EPI, STRIDED>(a, st);
  if (a.M <= 64) return pw_launch_cfg<T, VEC, 2, 2, PRO, EPI, STRIDED>(a, st);
  if (a.M <= 96) return pw_launch_cfg<T, VEC, 3, 2, PRO, EPI, STRIDED>(a, st);
  return pw_launch_cfg<T, VEC, 4, 2, PRO, EPI, STRIDED>(a, st);
}

template <typename T, int PRO, int EPI>
static int pw_launch_vec(PwGemmArgs& a, int vec, hipStream_t st) {
  constexpr int FULL = 16 / sizeof(T);
  if (a.stride > 1) {
    if constexpr (PRO == PRO_NONE && EPI == EPI_STATS)
      return pw_launch_tile<T, 1, PRO, EPI, true>(a, st);
    else {
      x3d_set_error("pw: strided gather only in forward");
      return X3D_ERR_INVALID;
    }
  }
  if (vec >= FULL) return pw_launch_tile<T, FULL, PRO, EPI, false>(a, st);
  return pw_launch_tile<T, 1, PRO, EPI, false>(a, st);
}

