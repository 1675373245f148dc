// Exact-fp32 pointwise GEMM with the weights RESIDENT in LDS (fp32 storage: BASELINE config 2, X3D-S 32 x 13 x 160^2).
//
// pw_gemm.h's kernel stages a [K chunk][BM] weight block AND a [K chunk][64 points] activation block for every 64-point
// tile and chunk, synchronously (stage -> barrier -> MFMA -> barrier), four waves per workgroup: the stage-4/5 layers ran at
// 0.4-1.1 TB/s (192 -> 432 on 13x5x5: 125 us for 52 MB and 1.7 GFLOP -- 9 % of the fp32 matrix-core rate).  Here
//   * a PERSISTENT 512-thread workgroup loads its [BM][K] weight block into LDS once (odd pitch: conflict-free b32 reads of
//     the v_mfma_f32_32x32x2_f32 A operand, one f32 per lane) and walks its share of the point tiles;
//   * activations stream in chunks of 32 k-rows x BN points through TWO LDS buffers: chunk c+1 is loaded into registers while
//     chunk c is multiplied, the prologue (folded BN / SE gate / activation, BN-backward) runs at the commit, once per element
//     -- one barrier per chunk;
//   * eight waves, one or two 32x32 output tiles each; the epilogue is pw_gemm.h's (the point index sits on the lane: every
//     store / load of the epilogue is 128 contiguous bytes per half-wave).
// Products and sums are the same exact fp32 operations; only the k order inside an output element is unchanged too
// (ascending), so results agree with pw_gemm_kernel's to the last bit wherever the tiling does not change the atomics' order.
#pragma once
#include "pw_gemm.h"

#define F32R_THREADS 512
#define F32R_KC 32
#ifndef F32R_EXP
#define F32R_EXP 0    // timing experiments (tools/ab_f32r_parts.sh; results WRONG unless 0): 1 no MFMA, 2 no weight load, 4 no epilogue stores, 8 no activation loads, 16 no statistics atomics, 32 no commit, 64 no epilogue
#endif

__host__ __device__ static inline int f32r_wpitch(int K) { return (((K + F32R_KC - 1) / F32R_KC) * F32R_KC) | 1; }
static inline size_t f32r_lds_bytes(int K, int MT, int NT) {
  const int BM = MT * 32, BN = NT * 32, Kp = ((K + F32R_KC - 1) / F32R_KC) * F32R_KC;
  return ((size_t)BM * f32r_wpitch(K) + 3 + (size_t)2 * F32R_KC * BN + (size_t)Kp * 4 + (size_t)BM * 4) * sizeof(float) + 16;
}

template <int VEC, int MT, int NT, int PRO, int EPI>
__global__ __launch_bounds__(F32R_THREADS) void pw_f32r_kernel(const PwGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef float T;
  constexpr int BM = MT * 32, BN = NT * 32, KC = F32R_KC, NW = F32R_THREADS / 64;
  constexpr int NTILE = MT * NT, TPW = (NTILE + NW - 1) / NW;
  constexpr bool HAS_SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  constexpr bool TWO = (PRO == PRO_BNBWD) || (PRO == PRO_TAIL);    // a second streamed tensor (a.x2)
  constexpr bool SIDE = (PRO == PRO_TAIL) || (PRO == PRO_AFFST);   // the activated input is also stored (a.ystore): the folded tail / stem BN
  constexpr int VPR = BN / VEC;                              // staging vectors per k row
  constexpr int NXV = KC * VPR / F32R_THREADS;               // ... per thread and chunk
  static_assert(KC * VPR % F32R_THREADS == 0, "chunk must divide over the workgroup");
  const int WP = f32r_wpitch(a.K);
  const int nchunks = (a.K + KC - 1) / KC, Kp = nchunks * KC;
  float* Ws = smem;                                          // [BM][WP]
  float* Xs = smem + ((BM * WP + 3) & ~3);                   // [2][KC][BN]
  float* Pk = Xs + 2 * KC * BN;                              // [Kp][4]  prologue rows
  float* Em = Pk + Kp * 4;                                   // [BM][4]  epilogue rows

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * BM;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  if (tile_begin >= tile_end) return;

  // ---- the weight block, once: Ws[m][k] = w(k, m0 + m), zero padding; walked along the contiguous axis of w, four elements
  // per load where rows allow it (a 64 x 432 block is 110 KB: in scalar rounds of 8 loads its latency was ~15 us per workgroup)
  if (!(F32R_EXP & 2)) {
    const int total = BM * Kp;
    const bool vec4 = a.wsk == 1 ? ((a.K & 3) == 0 && (a.wsm & 3) == 0) : (a.wsm == 1 && (a.M & 3) == 0 && (a.wsk & 3) == 0);
    if (vec4 && (((uintptr_t)a.w) & 15) == 0) {
      constexpr int UW = 4;
      for (int base = 0; base < total / 4; base += F32R_THREADS * UW) {
        f32x4 wv[UW];
        int dk[UW], dm[UW];
#pragma unroll
        for (int u = 0; u < UW; u++) {
          const int i = (base + u * F32R_THREADS + tid) * 4;
          int k, m;
          if (a.wsk == 1) { m = i / Kp; k = i - m * Kp; }
          else { k = i / BM; m = i - k * BM; }
          const bool in = i < total;
          dk[u] = in ? k : -1; dm[u] = m;
          wv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const bool ok = in && (a.wsk == 1 ? (k < a.K && m0 + m < a.M) : (k < a.K && m0 + m < a.M));   // (groups of 4 never straddle K / M: both are multiples of 4)
          if (ok) wv[u] = *(const f32x4*)(a.w + (long long)k * a.wsk + (long long)(m0 + m) * a.wsm);
        }
#pragma unroll
        for (int u = 0; u < UW; u++) {
          if (dk[u] < 0) continue;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (a.wsk == 1) Ws[dm[u] * WP + dk[u] + e] = wv[u][e];
            else Ws[(dm[u] + e) * WP + dk[u]] = wv[u][e];
          }
        }
      }
    } else {
      constexpr int UW = 8;
      for (int base = 0; base < total; base += F32R_THREADS * UW) {
        float wv[UW];
        int dst[UW];
#pragma unroll
        for (int u = 0; u < UW; u++) {
          const int i = base + u * F32R_THREADS + tid;
          int k, m;
          if (a.wsk == 1) { m = i / Kp; k = i - m * Kp; }
          else { k = i / BM; m = i - k * BM; }
          const bool in = i < total;
          dst[u] = in ? m * WP + k : -1;
          wv[u] = (in && k < a.K && m0 + m < a.M) ? a.w[(long long)k * a.wsk + (long long)(m0 + m) * a.wsm] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UW; u++) if (dst[u] >= 0) Ws[dst[u]] = wv[u];
      }
    }
  }
  auto fill_tables = [&](int n) {
    if constexpr (PRO != PRO_NONE) {
      for (int k = tid; k < Kp; k += F32R_THREADS) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (k < a.K) {
          if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFST) {
            c0 = a.coef[k * 2]; c1 = a.coef[k * 2 + 1];
            c2 = a.gate ? a.gate[(long long)n * a.K + k] : 1.0f;
          } else if constexpr (PRO == PRO_TAIL) {   // s_c * x + (s_r | 1) * x2 + (t_c + t_r | 0), then ReLU (pw_gemm_bf16.h)
            c0 = a.coef[k * 2]; c1 = a.coef2 ? a.coef2[k * 2] : 1.0f;
            c2 = a.coef[k * 2 + 1] + (a.coef2 ? a.coef2[k * 2 + 1] : 0.f);
          } else {
            c0 = a.coef[k * 4]; c1 = a.coef[k * 4 + 1]; c2 = a.coef[k * 4 + 2];
          }
        }
        Pk[k * 4] = c0; Pk[k * 4 + 1] = c1; Pk[k * 4 + 2] = c2;
      }
    }
    if constexpr (EPI == X3D_EPI_SWISH_BWD) {
      for (int m = tid; m < BM; m += F32R_THREADS) {
        const int gm = m0 + m;
        const bool ok = gm < a.M;
        Em[m * 4] = ok ? a.b_ss[gm * 2] : 0.f;
        Em[m * 4 + 1] = ok ? a.b_ss[gm * 2 + 1] : 0.f;
        Em[m * 4 + 2] = (ok && a.egate) ? a.egate[(long long)n * a.M + gm] : 1.0f;
      }
    }
    if constexpr (EPI == EPI_BNADD) {
      for (int m = tid; m < BM; m += F32R_THREADS) {
        float c0, c1, c2;
        bnadd_coef(a, m0 + m, m0 + m < a.M, c0, c1, c2);
        Em[m * 4] = c0; Em[m * 4 + 1] = c1; Em[m * 4 + 2] = c2;
      }
    }
  };

  // ---- activation chunks: global -> registers (one chunk ahead) -> prologue -> LDS buffer (step & 1)
  // (measured: a second register set -- two chunks in flight, the step loop unrolled by two -- changed nothing on the layers it
  // was meant for (432 -> 192 on 13x5x5: 107 -> 110 us) and the doubled epilogue code made the swish' instantiations 2-4x
  // slower: one set)
  float xr[NXV][VEC], yr[TWO ? NXV : 1][VEC];
  auto issue = [&](int tile, int kc) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NXV; i++) {
      const int v = tid + i * F32R_THREADS;
      const int k = kc * KC + v / VPR;
      const long long p = p0 + (long long)(v % VPR) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; e++) { xr[i][e] = 0.f; if constexpr (TWO) yr[i][e] = 0.f; }
      if (!(F32R_EXP & 8) && k < a.K && p < a.P) {
        if (VEC == 1 || p + VEC <= a.P) {
          // (rows of P % 4 != 0 points -- 13 frames of 5 x 5 -- start at any 4-byte address: the compute queues run in unaligned
          // access mode, so the 16-byte loads stay; the scalar staging form took twice the time per chunk)
          pw_load_raw<T, VEC, TWO ? PRO_BNBWD : PRO_NONE, false>(a, n, k, p, xr[i], yr[TWO ? i : 0]);
        } else {            // the row ends inside this vector: its elements one by one
          const long long o = ((long long)n * a.K + k) * a.Pin + p;
#pragma unroll
          for (int e = 0; e < VEC; e++) {
            if (p + e < a.P) {
              xr[i][e] = ((const T*)a.x)[o + e];
              if constexpr (TWO) yr[i][e] = ((const T*)a.x2)[o + e];
            }
          }
        }
      }
    }
  };
  auto commit = [&](int tile, int kc, float* buf) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NXV; i++) {
      const int v = tid + i * F32R_THREADS;
      const int kl = v / VPR, pv = v % VPR;
      const int k = kc * KC + kl;
      const long long p = p0 + (long long)pv * VEC;
      float val[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) val[e] = xr[i][e];
      if (k < a.K && p < a.P) {
        if constexpr (PRO == PRO_TAIL) {
          const float* ck = Pk + k * 4;
#pragma unroll
          for (int e = 0; e < VEC; e++) val[e] = fmaxf(ck[0] * val[e] + ck[1] * yr[TWO ? i : 0][e] + ck[2], 0.f);
        } else {
          pw_prologue<VEC, PRO == PRO_AFFST ? PRO_AFFINE : PRO>(a, Pk + k * 4, val, yr[TWO ? i : 0]);
        }
        if constexpr (SIDE) {     // y of the block below (the stem), kept for its other readers: written by the first row group
          if (blockIdx.y == 0) {
            T* yd = (T*)a.ystore + ((long long)n * a.K + k) * a.Pin + p;
            if (VEC == 1 || p + VEC <= a.P) VecIO<T, VEC>::store(yd, val);
            else {
#pragma unroll
              for (int e = 0; e < VEC; e++) if (p + e < a.P) yd[e] = val[e];
            }
          }
        }
        if (VEC > 1 && p + VEC > a.P) {
#pragma unroll
          for (int e = 0; e < VEC; e++) if (p + e >= a.P) val[e] = 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) val[e] = 0.f;     // the prologue constants must not leak into the padding
      }
      VecIO<float, VEC>::store(&buf[kl * BN + pv * VEC], val);
    }
  };

  f32x16 acc[TPW];
  float st1[HAS_SUMS ? TPW : 1][16], st2[HAS_SUMS ? TPW : 1][16];
  if constexpr (HAS_SUMS) {
#pragma unroll
    for (int s = 0; s < TPW; s++)
#pragma unroll
      for (int j = 0; j < 16; j++) { st1[s][j] = 0.f; st2[s][j] = 0.f; }
  }
  // SWISH_BWD: per-(sample, row) sums, flushed when the workgroup moves to another sample
  auto flush_nc = [&](int n) {
    if constexpr (EPI == X3D_EPI_SWISH_BWD) {
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        const int id = wid + NW * s;
        if (id < NTILE) {
          const int mt = id / NT;
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const float s1 = half_wave_sum_hi(st1[s][j]);
            const float s2 = half_wave_sum_hi(st2[s][j]);
            const int m = m0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
            if (r == 16 && m < a.M) {
              double* d = a.nc_sums + ((long long)n * a.M + m) * 2;
              atomic_add_d(d, (double)s1);
              atomic_add_d(d + 1, (double)s2);
            }
            st1[s][j] = 0.f; st2[s][j] = 0.f;
          }
        }
      }
    }
  };

  int n_cur = tile_begin / tiles_per_n;
  fill_tables(n_cur);
  issue(tile_begin, 0);
  __syncthreads();                      // weights and tables in place
  int step = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    if (n != n_cur) {                   // (the registers hold RAW chunk 0 of this tile: the tables only matter from its commit on)
      flush_nc(n_cur);
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFST || EPI == X3D_EPI_SWISH_BWD) {
        __syncthreads();                // every commit / epilogue of the previous sample has read its rows
        fill_tables(n);
        __syncthreads();
      }
      n_cur = n;
    }
#pragma unroll
    for (int s = 0; s < TPW; s++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

    for (int kc = 0; kc < nchunks; ++kc, ++step) {
      float* buf = Xs + (step & 1) * KC * BN;
      if (!(F32R_EXP & 32)) commit(tile, kc, buf);
      __syncthreads();                  // chunk visible; every wave is past the MFMAs of the chunk that used the other buffer
      {                                 // next chunk (of this tile or of the next one) flies under the MFMAs
        int nt_ = tile, nk = kc + 1;
        if (nk == nchunks) { nk = 0; nt_ = tile + 1; }
        if (nt_ < tile_end) issue(nt_, nk);
      }
      const int k0 = kc * KC;
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        const int id = wid + NW * s;
        if (id < NTILE) {
          const int mt = id / NT, nt = id - mt * NT;
          const float* wp = Ws + (mt * 32 + r) * WP + k0 + half;
          const float* xp = buf + half * BN + nt * 32 + r;
#pragma unroll 8
          for (int kk = (F32R_EXP & 1) ? KC - 2 : 0; kk < KC; kk += 2) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[kk], xp[kk * BN], acc[s], 0, 0, 0);
        }
      }
    }

    // ---- epilogue (pw_gemm.h's): D[row][col]: col = lane & 31 (point), row = (j & 3) + 8 (j >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + NW * s;
      if ((F32R_EXP & 64) && acc[s][0] != 12345.f) continue;
      if (id < NTILE) {
        const int mt = id / NT, nt = id - mt * NT;
        const long long p = p0 + nt * 32 + r;
        constexpr bool EPL = (EPI == X3D_EPI_ADD) || (EPI == X3D_EPI_SWISH_BWD) || (EPI == EPI_BNADD);
        float eop[EPL ? 16 : 1];
        if constexpr (EPL) {
          const T* esrc = (const T*)(EPI == X3D_EPI_SWISH_BWD ? a.braw : a.add);
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const int m = m0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
            eop[j] = ((m < a.M) && (p < a.P) && esrc) ? esrc[((long long)n * a.M + m) * a.P + p] : 0.f;
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
              if (!(F32R_EXP & 4) || val == 12345.f) ((T*)a.y)[o] = val;
              st1[s][j] += val;
              st2[s][j] += val * val;
            }
          } else if constexpr (EPI == X3D_EPI_STORE) {
            if (ok) ((T*)a.y)[o] = val;
          } else if constexpr (EPI == EPI_BNADD) {
            if (ok) {
              const float* em = Em + (m - m0) * 4;
              float v = em[0] * val + em[1] + em[2] * eop[j];
              if (a.eact == X3D_ACT_RELU) v = fmaxf(v, 0.f);
              ((T*)a.y)[o] = v;
            }
          } else if constexpr (EPI == X3D_EPI_ADD) {
            if (ok) ((T*)a.y)[o] = val + eop[j];
          } else if constexpr (EPI == X3D_EPI_ADD_STRIDED) {
            if (ok) {
              const int hw = a.eH * a.eW;
              const int t = (int)p / hw;
              const int rem = (int)p - t * hw;
              const int h = rem / a.eW, w = rem - h * a.eW;
              if (((h | w) & 1) == 0) {
                const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
                const long long T_ = a.P / hw;
                const long long oa = ((((long long)n * a.M + m) * T_ + t) * Hh + (h >> 1)) * Wh + (w >> 1);
                val += ((const T*)a.add)[oa];
              }
              ((T*)a.y)[o] = val;
            }
          } else if constexpr (EPI == X3D_EPI_SWISH_BWD) {
            if (ok) {
              const float b = eop[j];
              const float* em = Em + (m - m0) * 4;
              const float u = em[0] * b + em[1];
              const float g = em[2];
              const float dv = val * swish_grad_(u * g);
              ((T*)a.y)[o] = dv;
              st1[s][j] += dv;
              st2[s][j] += dv * b;
            }
          }
        }
      }
    }
  }
  flush_nc(n_cur);

  if constexpr (EPI == EPI_STATS) {
    // the NT waves of a row block hold partial sums of the same channels: added up in LDS (in double, as the atomics would), then
    // ONE pair of atomics per channel and workgroup (the per-wave flush was 4 x the atomics on 32 copies of a few hundred
    // addresses: 10-17 us of a 55-75 us launch, profiles/r06_ab_f32r_parts.txt)
    float* red = Xs;                    // [NT][BM][2], free once every wave is past its last MFMA
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + NW * s;
      if (id < NTILE) {
        const int mt = id / NT, nt = id - mt * NT;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const float s1 = half_wave_sum_hi(st1[s][j]);
          const float s2 = half_wave_sum_hi(st2[s][j]);
          const int ml = mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
          if (r == 16) { red[(nt * BM + ml) * 2] = s1; red[(nt * BM + ml) * 2 + 1] = s2; }
        }
      }
    }
    __syncthreads();
    if (tid < BM * 2 && a.stats && !(F32R_EXP & 16)) {
      const int ml = tid >> 1, q = tid & 1, m = m0 + ml;
      if (m < a.M) {
        double t = 0.0;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) t += (double)red[(nt * BM + ml) * 2 + q];
        atomic_add_d(&stats_replica(a.stats, a.M, blockIdx.x)[m * 2 + q], t);
      }
    }
  }
}

// X3D_PW_F32R=0: A/B hook (the per-tile staging kernel of pw_gemm.h everywhere)
static inline bool f32r_enabled() { return x3d_env_int("X3D_PW_F32R", 1) != 0; }

// tile shape: as many 32-row blocks of the output as fit LDS next to the two activation buffers
// (max_mt: the epilogues that keep per-element sums -- statistics, swish' -- hold 64 more registers per two tiles and spill at
// two tiles per wave: they take row groups of 64)
static inline bool f32r_shape(const PwGemmArgs& a, int max_mt, int* MT, int* NT) {
  int mt = ceil_div(a.M, 32);
  if (mt > 4) mt = 4;
  if (a.M > 64 && a.M <= 96) mt = 3;
  if (mt > max_mt) mt = max_mt;
  for (; mt >= 1; mt--) {
    const int nt = mt == 1 ? 8 : 4;
    if (f32r_lds_bytes(a.K, mt, nt) <= 160 * 1024) { *MT = mt; *NT = nt; return true; }   // (K = 432: 64 rows x 449 + the two chunk buffers = 152 KB)
  }
  return false;
}

template <int VEC, int MT, int NT, int PRO, int EPI>
static int f32r_launch_cfg(PwGemmArgs& a, hipStream_t st) {
  constexpr int BM = MT * 32, BN = NT * 32;
  const size_t lds = f32r_lds_bytes(a.K, MT, NT);
  X3D_DESCRIBE("pw_f32r_kernel<%d, %d, %d, %d, %d>", VEC, MT, NT, PRO, EPI);
  auto kern = pw_f32r_kernel<VEC, MT, NT, PRO, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  static size_t occ_lds[8];
  static int occ_slots[8], occ_n = 0;
  int slots = 0;
  for (int i = 0; i < occ_n; i++) if (occ_lds[i] == lds) slots = occ_slots[i];
  if (slots == 0) {
    int nb = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, F32R_THREADS, lds) != hipSuccess || nb < 1) nb = 1;
    slots = nb * cus;
    if (occ_n < 8) { occ_lds[occ_n] = lds; occ_slots[occ_n] = slots; occ_n++; }
  }
  const int gy = ceil_div(a.M, BM);
  const long long total_tiles = ceil_div_ll(a.P, BN) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_f32r: too many tiles");
  long long per_group = slots / gy;                    // one round of persistent workgroups over all row groups
  if (per_group < 1) per_group = 1;
  long long tpb = ceil_div_ll(total_tiles, per_group);
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = (int)tpb;
  const long long gx = ceil_div_ll(total_tiles, tpb);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(F32R_THREADS), lds, st, a);
  X3D_LAUNCH_CHECK("pw_f32r");
  return X3D_OK;
}

// returns -1 when the launch is not covered (the caller falls back to pw_gemm_kernel)
template <int PRO, int EPI>
static int f32r_try(PwGemmArgs& a, int vec, hipStream_t st) {
  if (!f32r_enabled() || a.stride != 1 || a.P >= (1ll << 31)) return -1;
  int MT = 0, NT = 0;
  constexpr bool SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  if (!f32r_shape(a, SUMS ? 2 : 4, &MT, &NT)) return -1;
#define F32R_CASE(V_, M_, N_) if (MT == M_ && NT == N_) return f32r_launch_cfg<V_, M_, N_, PRO, EPI>(a, st);
  // four-element staging vectors whatever the row length (P % 4 != 0: unaligned loads, element tails); pointers 4-byte aligned
  (void)vec;
  if (a.P < 4) return -1;
  F32R_CASE(4, 1, 8) F32R_CASE(4, 2, 4)
  if constexpr (!SUMS) { F32R_CASE(4, 3, 4) F32R_CASE(4, 4, 4) }
#undef F32R_CASE
  return -1;
}
