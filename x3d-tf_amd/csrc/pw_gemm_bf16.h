// Pointwise-conv GEMMs for bf16 activation storage on v_mfma_f32_32x32x16_bf16.
//
//   D[M rows][N = points] = W[M][K] * X[K][N]          (fwd: M=Cout,K=Cin;  dgrad: M=Cin,K=Cout)
//
// These GEMMs have K <= 432 and are HBM-bound by a wide margin (the MFMA work of a 128-point tile is
// ~100-400 cycles against ~2000 cycles of HBM time), so the kernel is organised around the memory system:
//  * A operand (weights): staged once per workgroup in LDS as [m][k] bf16 (fp32 master weights converted
//    on the fly), read with ds_read_b128 (row pitch K+8 elements: the 16 lanes of a b128 group hit 16
//    distinct slots).
//  * B operand (activations): NCTHW makes the POINTS contiguous but the MFMA wants eight CHANNELS of one
//    point per lane.  The tile is staged [k][points] exactly as it lies in HBM (16-byte coalesced loads,
//    prologue transform in fp32, ds_write_b128) and read back transposed with ds_read_b64_tr_b16, the
//    gfx950 hardware transpose (pitch 2*BN+64 bytes: the 4 rows of a half-wave sit on disjoint banks).
//  * software pipeline: the global loads of the NEXT K-chunk / point tile are issued into registers before
//    the MFMAs and the epilogue of the current one, so HBM latency hides behind them.
//  * epilogue through LDS: the 32x32 accumulators (point index on the lane) are written to an fp32 LDS tile
//    and re-read row-wise, so every global access of the epilogue (output store, residual add, raw
//    depthwise output for swish') is a coalesced 16-byte row access, and the per-channel statistics are
//    accumulated by the thread that owns the row (16 registers instead of 2x16 per accumulator tile).
#pragma once
#include <stdlib.h>

#include "pw_gemm.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define PWB_BN 128
#define PWB_KCH 64
#define PWB_XP (PWB_BN + 32)   // X pitch (elements)
#define PWB_OP (PWB_BN + 4)    // output-tile pitch (floats)

// RAG: rows of P % 8 != 0 points (common.h, pw_ragged_rows).  A separate instantiation: the element accesses of a row's
// last vector sit in branches, and with them in the loop the compiler's vmcnt counts of the ALIGNED layers turned
// conservative (216 -> 96 @ 14x14 forward 54 -> 87 us) -- so the aligned instantiations do not contain them.
template <typename H, int VEC, int MT, int PRO, int EPI, int STRIDED, int OVEC, bool RAG = false>
__global__ __launch_bounds__(256) void pw_gemm_bf16_kernel(const PwGemmArgs a) {
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4; typedef typename HV<H>::x2 hx2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BM = MT * 32, BN = PWB_BN, XP = PWB_XP, OP = PWB_OP, KCH = PWB_KCH;
  constexpr bool HAS_SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  // buffer stores with a static count per tile (see the epilogue); M*P*2 < 2^31 bytes is checked by the host
  constexpr bool BSTORE = (OVEC == 8);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  constexpr int VPR = BN / VEC;                       // staging vectors per row
  constexpr int NSV = (KCH * VPR + 255) / 256;        // staging vectors per thread per chunk
  constexpr int ROWS_PT = BM / 16;                    // output rows owned per thread in the epilogue passes
  // the fp32 output tile goes through LDS in slabs of SLAB 32-row tiles, so a tall panel (MT up to 7: every
  // output channel of a stage-4 layer from ONE staging + prologue pass over the activations) costs no extra LDS
  constexpr int SLAB = (MT <= 2) ? MT : 1, NSLAB = MT / SLAB, ROWS_SL = 2 * SLAB;
  const int Kp = a.KC;                                // K rounded up to 16
  const int WP = Kp + 8;
  // the activation chunk and the fp32 output tile are never live together: they share one LDS region
  // (a 34 KB saving that keeps two workgroups per CU up to K ~ 300)
  constexpr size_t XO_BYTES = ((size_t)KCH * XP * 2 > (size_t)SLAB * 32 * OP * 4) ? (size_t)KCH * XP * 2 : (size_t)SLAB * 32 * OP * 4;
  H* Xs = (H*)smem_raw;                         // [KCH][XP]
  float* Os = (float*)smem_raw;                       // [SLAB * 32][OP]   (aliases Xs)
  H* Ws = (H*)(smem_raw + XO_BYTES);            // [BM][WP]
  // per-row prologue coefficients in LDS ([Kp][4] floats, zero for padded rows): AFFINE {s*gate, t*gate}, BNBWD
  // {A, B, C}.  Read from global inside the prologue they cost an exposed L2 round trip per chunk.
  float* Cs = (float*)(smem_raw + XO_BYTES + (size_t)BM * (a.KC + 8) * 2);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * BM;
  // tiles are numbered across samples (tile -> sample n = tile / tiles_per_n), so a workgroup amortises its
  // weight panel over `tiles_per_block` tiles even when one sample has only a handful (P = 784 at stage 5)
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  const int nkc = (Kp + KCH - 1) / KCH;

  // ---- resident weight panel: Ws[m][k] = W(k, m) as bf16, zero padded.  The fp32 master weights are read
  // along their contiguous axis with the widest aligned vector (4/2/1 floats).
  if (a.wp) {
    // packed panel: the LDS image itself, rows m0.. of pitch WP -> a flat 16-byte copy (rows past the panel: zero)
    const hx8* src = (const hx8*)((const H*)a.wp + (long long)m0 * WP);
    const int nvec = BM * WP / 8;
    const int lim = max(0, min(BM, a.wp_rows - m0)) * WP / 8;
    hx8 zero;
#pragma unroll
    for (int e = 0; e < 8; e++) zero[e] = (H)0.f;
#pragma unroll 4
    for (int i = tid; i < nvec; i += 256) ((hx8*)Ws)[i] = i < lim ? src[i] : zero;
  } else {
    const int wv = a.wvec;
    if (a.wsk == 1) {            // forward: source rows are k-contiguous -> one LDS row segment per vector
      const int kv = Kp / wv;    // Kp % 4 == 0
      for (int i = tid; i < BM * kv; i += 256) {
        const int m = i / kv, k = (i - m * kv) * wv;
        const int gm = m0 + m;
        const float* src = a.w + (long long)gm * a.wsm + k;
        H* dst = &Ws[m * WP + k];
        for (int e = 0; e < wv; e += 1) dst[e] = (H)0.f;
        if (gm < a.M && k < a.K) {
          if (wv == 4) { const f32x4 v = *(const f32x4*)src; dst[0] = (H)v[0]; dst[1] = (H)v[1]; dst[2] = (H)v[2]; dst[3] = (H)v[3]; }
          else if (wv == 2) { const float2 v = *(const float2*)src; dst[0] = (H)v.x; dst[1] = (H)v.y; }
          else dst[0] = (H)src[0];
        }
      }
    } else {                     // dgrad: source rows are m-contiguous -> a vector scatters over wv LDS rows
      const int mv = BM / wv;
      for (int i = tid; i < Kp * mv; i += 256) {
        const int k = i / mv, m = (i - k * mv) * wv;
        const int gm = m0 + m;
        const float* src = a.w + (long long)k * a.wsk + gm;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < a.K) {
          if (wv == 4 && gm + 3 < a.M) { const f32x4 t = *(const f32x4*)src; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
          else for (int e = 0; e < wv; e++) if (gm + e < a.M) v[e] = src[e];
        }
        for (int e = 0; e < wv; e++) Ws[(m + e) * WP + k] = (H)v[e];
      }
    }
  }

  // AFFINE rows are 2 floats (table always in LDS), BNBWD rows 4 floats (in LDS unless it costs a resident workgroup: K >= 320)
  constexpr bool AFF = (PRO == PRO_AFFINE) || (PRO == PRO_AFFST);   // one-tensor affine prologue (AFFST: + the side store)
  constexpr int CSW = AFF ? 2 : 4;   // (BNBWD, TAIL: three coefficients per row)
  const bool use_cs = AFF || a.K < 320;
  auto fill_coef = [&](int n) {
    if (!use_cs) return;
    if constexpr (PRO != PRO_NONE) {
      for (int k = tid; k < Kp; k += 256) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        if (k < a.K) {
          if constexpr (AFF) {
            const float g = a.gate ? a.gate[(long long)n * a.K + k] : 1.0f;
            c[0] = a.coef[k * 2] * g;       // (s*x + t) * g
            c[1] = a.coef[k * 2 + 1] * g;
          } else if constexpr (PRO == PRO_TAIL) {   // s_c*x + (s_r|1)*x2 + (t_c + t_r|0)
            c[0] = a.coef[k * 2]; c[1] = a.coef2 ? a.coef2[k * 2] : 1.0f;
            c[2] = a.coef[k * 2 + 1] + (a.coef2 ? a.coef2[k * 2 + 1] : 0.f);
          } else {
            c[0] = a.coef[k * 4]; c[1] = a.coef[k * 4 + 1]; c[2] = a.coef[k * 4 + 2];
          }
        }
        if constexpr (CSW == 2) *(float2*)&Cs[k * 2] = make_float2(c[0], c[1]);
        else *(f32x4*)&Cs[k * 4] = c;
      }
    }
  };
  if (tile_begin < tile_end) fill_coef(tile_begin / tiles_per_n);

  // ---- register-staged prefetch of one [kc][BN] chunk
  constexpr bool TWO = (PRO == PRO_BNBWD) || (PRO == PRO_TAIL);   // a second streamed tensor (a.x2)
  hx8 xr[NSV], yr[(TWO || STRIDED) ? NSV : 1];   // raw loads (VEC == 8); scalar path uses xs1[]
  float xs1[(VEC == 1) ? NSV : 1], ys1[(VEC == 1 && TWO) ? NSV : 1];
  auto issue_loads = [&](int tile, int kc_idx) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    const int k0 = kc_idx * KCH;
    const int kc = min(KCH, Kp - k0);
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * 256;
      const int k = v / VPR, pv = v - k * VPR;
      const int gk = k0 + k;
      const long long p = p0 + (long long)pv * VEC;
      const bool ok = (k < kc) && (gk < a.K) && (p < a.P);
      if constexpr (VEC == 8) {
        hx8 z;
#pragma unroll
        for (int e = 0; e < 8; e++) z[e] = (H)0.f;
        xr[i] = z;
        if constexpr (TWO) yr[i] = z;
        if constexpr (STRIDED) yr[i] = z;
        if (ok) {
          if constexpr (STRIDED) {
            // strided shortcut: gather the even input elements, STRIDED = outputs per aligned load (common.h)
            strided_gather16<STRIDED>((const T*)a.x + ((long long)n * a.K + gk) * a.Pin, p, a.H, a.W, a.Ho, a.Wo, xr[i], yr[i],
                                      RAG ? (int)min((long long)8, a.P - p) : 8);
          } else {
            // row (tid>>4) + 16*i of the chunk, 8 points at column 8*(tid&15): one base, stride 16 rows
            const long long o = ((long long)n * a.K + k0 + (tid >> 4)) * a.Pin + p0 + (tid & 15) * 8 + (long long)i * 16 * a.Pin;
            const long long left = RAG ? a.P - p : 8;          // RAG: the row may end inside its last vector
            if (left >= 8) {
              xr[i] = *(const hx8*)((const T*)a.x + o);
              if constexpr (TWO) yr[i] = *(const hx8*)((const T*)a.x2 + o);
            } else {
              xr[i] = load8_ragged<T, hx8>((const T*)a.x + o, (int)left);
              if constexpr (TWO) yr[i] = load8_ragged<T, hx8>((const T*)a.x2 + o, (int)left);
            }
          }
        }
      } else {
        xs1[i] = 0.f;
        if constexpr (TWO) ys1[i] = 0.f;
        if (ok) {
          long long src = p;
          if constexpr (STRIDED) {
            const int hw = a.Ho * a.Wo;
            const int t = (int)p / hw;
            const int rem = (int)p - t * hw;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            src = ((long long)t * a.H + (long long)ho * a.stride) * a.W + (long long)wo * a.stride;
          }
          const long long o = ((long long)n * a.K + gk) * a.Pin + src;
          xs1[i] = to_f<T>(((const T*)a.x)[o]);
          if constexpr (TWO) ys1[i] = to_f<T>(((const T*)a.x2)[o]);
        }
      }
    }
  };
  // transform (fp32) + write the prefetched chunk to LDS
  auto commit = [&](int tile, int kc_idx) {
    const int n = tile / tiles_per_n;
    const int k0 = kc_idx * KCH;
    const int kc = min(KCH, Kp - k0);
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * 256;
      const int k = v / VPR, pv = v - k * VPR;
      if (k >= kc) continue;
      const int gk = k0 + k;
      H* dst = &Xs[k * XP + pv * VEC];
      if constexpr (PRO == PRO_NONE) {
        if constexpr (VEC == 8 && STRIDED) {
          hx8 o;
#pragma unroll
          for (int e = 0; e < 4; e++) { o[e] = xr[i][2 * e]; o[4 + e] = yr[i][2 * e]; }
          *(hx8*)dst = o;
        } else if constexpr (VEC == 8) *(hx8*)dst = xr[i];
        else dst[0] = (H)xs1[i];
      } else {
        float val[VEC];
        if constexpr (VEC == 8) {
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] = (float)xr[i][e];
        } else {
          val[0] = xs1[i];
        }
        f32x4 cf;
        if (use_cs) {
          if constexpr (CSW == 2) { const float2 t2 = *(const float2*)&Cs[gk * 2]; cf[0] = t2.x; cf[1] = t2.y; cf[2] = 0.f; cf[3] = 0.f; }
          else cf = *(const f32x4*)&Cs[gk * 4];   // zeros for padded rows: they stay exactly zero
        } else {   // wide-K layers: the table would cost the third resident workgroup (48 -> 55 KB of LDS)
          const bool inb = gk < a.K;
          cf[3] = 0.f;
          if constexpr (AFF) {
            const float g = (inb && a.gate) ? a.gate[(long long)n * a.K + gk] : 1.0f;
            cf[0] = inb ? a.coef[gk * 2] * g : 0.f; cf[1] = inb ? a.coef[gk * 2 + 1] * g : 0.f; cf[2] = 0.f;
          } else if constexpr (PRO == PRO_TAIL) {
            cf[0] = inb ? a.coef[gk * 2] : 0.f; cf[1] = (inb && a.coef2) ? a.coef2[gk * 2] : (inb ? 1.0f : 0.f);
            cf[2] = inb ? a.coef[gk * 2 + 1] + (a.coef2 ? a.coef2[gk * 2 + 1] : 0.f) : 0.f;
          } else {
            cf[0] = inb ? a.coef[gk * 4] : 0.f; cf[1] = inb ? a.coef[gk * 4 + 1] : 0.f; cf[2] = inb ? a.coef[gk * 4 + 2] : 0.f;
          }
        }
        if constexpr (AFF) {
#pragma unroll
          for (int e = 0; e < VEC; e++) val[e] = cf[0] * val[e] + cf[1];
          act_vec<VEC>(val, a.act);
        } else {  // PRO_BNBWD, PRO_TAIL
#pragma unroll
          for (int e = 0; e < VEC; e++) {
            const float y2 = (VEC == 8) ? (float)yr[i][e] : ys1[i];
            val[e] = cf[0] * val[e] + cf[1] * y2 + cf[2];
            if constexpr (PRO == PRO_TAIL) val[e] = fmaxf(val[e], 0.f);
          }
        }
        VecIO<H, VEC>::store(dst, val);
        if constexpr (PRO == PRO_TAIL || PRO == PRO_AFFST) {   // the activated input IS the output y of the block below (the stem): kept for its other readers
          const int n_ = tile / tiles_per_n;
          const long long p_ = (long long)(tile - n_ * tiles_per_n) * BN + (long long)pv * VEC;
          if (blockIdx.y == 0 && gk < a.K && p_ < a.P) {
            T* yd = (T*)a.ystore + ((long long)n_ * a.K + gk) * a.Pin + p_;
            if constexpr (VEC == 8) {
              if (!RAG || a.P - p_ >= 8) VecIO<T, 8>::store(yd, val);
              else for (int e = 0; e < (int)(a.P - p_); e++) yd[e] = from_f<T>(val[e]);
            } else {
              yd[0] = from_f<T>(val[0]);
            }
          }
        }
      }
    }
  };

  float st1[HAS_SUMS ? ROWS_PT : 1], st2[HAS_SUMS ? ROWS_PT : 1];
  if constexpr (HAS_SUMS) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) { st1[i] = 0.f; st2[i] = 0.f; }
  }

  // transposed-read lane geometry
  const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_col = wid * 32 + 16 * (g16 & 1) + 4 * pp;    // this wave's point tile = wid
  const int tr_row = 8 * (g16 >> 1) + q;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

  f32x16 acc[MT];
  if (tile_begin < tile_end) issue_loads(tile_begin, 0);
  if constexpr (BSTORE) {
    // ROWS_PT discarded stores behind the first prefetch: the loop-carried path has exactly ROWS_PT stores behind
    // every prefetch, and the waitcnt pass only emits vmcnt(ROWS_PT) if the entry path looks the same
    __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((T*)a.y, 0, 0, 0x00020000);
    u32x4_ z0 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) __builtin_amdgcn_raw_buffer_store_b128(z0, r0, 0x80000000u, 0, 0);
  }

  // reduce the row sums over the 16 threads that share a row and publish them (fp64 atomics)
  auto flush_sums = [&](int n) {
    if constexpr (HAS_SUMS) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) {
        float s1 = st1[i], s2 = st2[i];
        s1 = row16_sum(s1);   // the 16 threads of a row are one DPP row
        s2 = row16_sum(s2);
        const int m = m0 + (tid >> 4) + 16 * i;
        if ((tid & 15) == 0 && m < a.M) {
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
        st1[i] = 0.f;
        st2[i] = 0.f;
      }
    }
  };

  // EPI_BNADD: the output rows' coefficients are the same for every tile of the workgroup
  float bna_s[EPI == EPI_BNADD ? ROWS_PT : 1], bna_t[EPI == EPI_BNADD ? ROWS_PT : 1], bna_g[EPI == EPI_BNADD ? ROWS_PT : 1];
  if constexpr (EPI == EPI_BNADD) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) {
      const int m = m0 + (tid >> 4) + 16 * i;
      bnadd_coef(a, m, m < a.M, bna_s[i], bna_t[i], bna_g[i]);
    }
  }

  int n_prev = tile_begin < tile_end ? tile_begin / tiles_per_n : 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    if constexpr (EPI == X3D_EPI_SWISH_BWD) {
      if (n != n_prev) flush_sums(n_prev);     // the sums are per (sample, channel)
    }
    if constexpr (AFF) {
      if (n != n_prev && a.gate) fill_coef(n);  // the SE gate is per sample; every reader of the old table is past its last barrier
    }
    n_prev = n;
#pragma unroll
    for (int s = 0; s < MT; s++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

    for (int kc_idx = 0; kc_idx < nkc; ++kc_idx) {
      const int k0 = kc_idx * KCH;
      const int kc = min(KCH, Kp - k0);
      __syncthreads();            // previous readers of Xs (and, at kc_idx 0, of Os) are done
      commit(tile, kc_idx);
      __syncthreads();
      // prefetch what comes next while the matrix cores and the epilogue run
      if (kc_idx + 1 < nkc) issue_loads(tile, kc_idx + 1);
      else if (tile + 1 < tile_end) issue_loads(tile + 1, 0);
      for (int kk = 0; kk < kc; kk += 16) {
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[(kk + tr_row) * XP + tr_col]));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[(kk + tr_row + 4) * XP + tr_col]));
        const s16x8 bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const hx8 bfrag = __builtin_bit_cast(hx8, bs);
#pragma unroll
        for (int s = 0; s < MT; s++) {
          const hx8 afrag = *(const hx8*)&Ws[(s * 32 + r) * WP + k0 + kk + 8 * half];
          acc[s] = mfma16<H>(afrag, bfrag, acc[s]);
        }
      }
    }

    // ---- accumulators -> LDS output slab (col = lane&31 -> point, rows in registers) -> row-wise epilogue pass:
    // thread owns rows (tid>>4) + 16*i of the slab, an 8-point column chunk (tid&15)
    const int oc = (tid & 15) * 8;
    __amdgpu_buffer_rsrc_t yrsrc;
    if constexpr (BSTORE)
      yrsrc = __builtin_amdgcn_make_buffer_rsrc((T*)a.y + (long long)n * a.M * a.P, 0, (int)((long long)a.M * a.P * 2), 0x00020000);
    // Every global load of the epilogue is issued HERE, before any store of this tile.  vmcnt retires in order
    // (stores included) and the compiler must assume vmcnt(0) around the conditional stores, so a load issued
    // after a store would wait for that store's write latency (once per row), and the first use of the next
    // tile's prefetched registers would wait for the last stores.  With the loads first, their wait also covers
    // the (older) prefetch, and nothing ever waits on a store.
    constexpr bool BNA_ = (EPI == EPI_BNADD);
    constexpr bool EPL8 = (OVEC == 8) && (EPI == X3D_EPI_ADD || EPI == X3D_EPI_SWISH_BWD || BNA_);
    constexpr bool EPL4 = (OVEC == 8) && (EPI == X3D_EPI_ADD_STRIDED);
    hx8 epl8[EPL8 ? ROWS_PT : 1];
    hx4 epl4[EPL4 ? ROWS_PT : 1];
    // the 8 points of a vector (p % 8 == 0) split into groups of egv points that stay inside one image row; each group's
    // even pixels receive egv / 2 contiguous half-resolution values.  egv = 8 / 4 / 2 by the row length; 0 (odd rows): scalar
    const int egv = !EPL4 ? 0 : ((a.eW & 7) == 0 ? 8 : ((a.eW & 3) == 0 ? 4 : ((a.eW & 1) == 0 ? 2 : 0)));
    const bool epl4_vec = egv != 0;
    constexpr bool SWB_ = (EPI == X3D_EPI_SWISH_BWD);
    // per-row BN_b scale/shift, SE gate (SWISH_BWD) | output scale, shift, scale of `add` (BNADD: loaded once, before the
    // tile loop -- a global load here sits BEHIND the next tile's prefetch and its wait is a vmcnt(0))
    float esb[(SWB_ || BNA_) ? ROWS_PT : 1], etb[(SWB_ || BNA_) ? ROWS_PT : 1], egt[(SWB_ || BNA_) ? ROWS_PT : 1];
    if constexpr (BNA_) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) { esb[i] = bna_s[i]; etb[i] = bna_t[i]; egt[i] = bna_g[i]; }
    }
    if constexpr (SWB_) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) {
        const int m = m0 + (tid >> 4) + 16 * i;
        const bool ok = m < a.M;
        esb[i] = ok ? a.b_ss[m * 2] : 0.f;
        etb[i] = ok ? a.b_ss[m * 2 + 1] : 0.f;
        egt[i] = (ok && a.egate) ? a.egate[(long long)n * a.M + m] : 1.0f;
      }
    }
    if constexpr (EPL8 || EPL4) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) {
        const int m = m0 + (tid >> 4) + 16 * i;
        const long long p = p0 + oc;
        if constexpr (EPL8) {
#pragma unroll
          for (int e = 0; e < 8; e++) epl8[i][e] = (H)0.f;
          if (m < a.M && p < a.P && (!BNA_ || a.add)) {
            const long long o = ((long long)n * a.M + m) * a.P + p;
            const T* src = (const T*)(EPI == X3D_EPI_SWISH_BWD ? a.braw : a.add) + o;
            if (!RAG || a.P - p >= 8) epl8[i] = *(const hx8*)src;
            else epl8[i] = load8_ragged<T, hx8>(src, (int)(a.P - p));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) epl4[i][e] = (H)0.f;
          if (epl4_vec && m < a.M && p < a.P) {
            const int hw = a.eH * a.eW;      // per-sample point counts fit 32 bits (host check): 32-bit divisions
            const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
            const int T_ = (int)a.P / hw;
            const T* abase = (const T*)a.add + ((long long)n * a.M + m) * T_ * Hh * Wh;
            if (egv == 8) {
              // the 8 points lie in one image row; on even rows the even ones receive 4 contiguous half-resolution values
              const int t = (int)p / hw;
              const int rem = (int)p - t * hw;
              const int h = rem / a.eW, w = rem - h * a.eW;
              if ((h & 1) == 0) epl4[i] = *(const hx4*)(abase + ((long long)t * Hh + (h >> 1)) * Wh + (w >> 1));
            } else if (egv == 4) {
#pragma unroll
              for (int gq = 0; gq < 2; gq++) {
                const int pe = (int)p + 4 * gq;
                if (RAG && pe >= (int)a.P) continue;
                const int t = pe / hw;
                const int rem = pe - t * hw;
                const int h = rem / a.eW, w = rem - h * a.eW;
                if ((h & 1) == 0) {
                  const hx2 v2 = *(const hx2*)(abase + ((long long)t * Hh + (h >> 1)) * Wh + (w >> 1));
                  epl4[i][2 * gq] = v2[0]; epl4[i][2 * gq + 1] = v2[1];
                }
              }
            } else {
#pragma unroll
              for (int gq = 0; gq < 4; gq++) {
                const int pe = (int)p + 2 * gq;
                if (RAG && pe >= (int)a.P) continue;
                const int t = pe / hw;
                const int rem = pe - t * hw;
                const int h = rem / a.eW, w = rem - h * a.eW;
                if ((h & 1) == 0) epl4[i][gq] = abase[((long long)t * Hh + (h >> 1)) * Wh + (w >> 1)];
              }
            }
          }
        }
      }
    }
    // (forward / plain-store epilogues have no loads to hoist; an explicit vmcnt(0) before their stores was measured
    // slower -- it exposes the just-issued prefetch of the next tile when K is one chunk)
#pragma unroll
    for (int sl = 0; sl < NSLAB; sl++) {
    __syncthreads();   // Os aliases Xs: every wave is done with the last chunk's fragments / the previous slab
#pragma unroll
    for (int s = 0; s < SLAB; s++)
#pragma unroll
      for (int j = 0; j < 16; j++)
        Os[(s * 32 + (j & 3) + 8 * (j >> 2) + 4 * half) * OP + wid * 32 + r] = acc[sl * SLAB + s][j];
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < ROWS_SL; ii++) {
      const int i = sl * ROWS_SL + ii;                 // index into the per-thread row sums
      const int row = (tid >> 4) + 16 * ii;            // row inside the slab
      const int m = m0 + sl * SLAB * 32 + row;
      const long long p = p0 + oc;
      // OVEC == 8: no branch around the row -- invalid rows / points run the arithmetic on zeros and their store
      // is dropped by the buffer bounds check, so the number of stores per tile is static and the compiler can
      // count them in s_waitcnt vmcnt(N) instead of waiting for every one of them (BSTORE below)
      const bool rvalid = (m < a.M) && (p < a.P);
      if constexpr (!BSTORE) { if (!rvalid) continue; }
      float val[8];
      {
        const f32x4 v0 = *(const f32x4*)&Os[row * OP + oc], v1 = *(const f32x4*)&Os[row * OP + oc + 4];
#pragma unroll
        for (int e = 0; e < 4; e++) { val[e] = v0[e]; val[4 + e] = v1[e]; }
      }
      const long long o = ((long long)n * a.M + m) * a.P + p;
      const int nvalid = (OVEC == 8 && !RAG) ? 8 : (int)min((long long)8, a.P - p);   // < 8: the row ends inside this vector
      if constexpr (EPI == X3D_EPI_ADD) {
        if constexpr (OVEC == 8) {
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] += (float)epl8[i][e];
        } else {
          for (int e = 0; e < nvalid; e++) val[e] += to_f<T>(((const T*)a.add)[o + e]);
        }
      } else if constexpr (BNA_) {
        float ad[8];
        if constexpr (OVEC == 8) {
#pragma unroll
          for (int e = 0; e < 8; e++) ad[e] = (float)epl8[i][e];
        } else {
          for (int e = 0; e < 8; e++) ad[e] = (e < nvalid && a.add) ? to_f<T>(((const T*)a.add)[o + e]) : 0.f;
        }
        const float lo = a.eact == X3D_ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
        for (int e = 0; e < 8; e++) val[e] = fmaxf(esb[i] * val[e] + etb[i] + egt[i] * ad[e], lo);
      } else if constexpr (EPI == X3D_EPI_ADD_STRIDED) {
        const int hw = a.eH * a.eW;      // per-sample point counts fit 32 bits (host check): 32-bit divisions
        const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
        const int T_ = (int)a.P / hw;
        if (epl4_vec) {
          if constexpr (EPL4) {   // loaded above (zeros on odd rows)
#pragma unroll
            for (int e = 0; e < 4; e++) val[2 * e] += (float)epl4[i][e];
          }
        } else if (m < a.M)   // (BSTORE runs the rows past M too: without this guard the last sample's padding rows read up to
                              //  31 rows * T * Hh * Wh elements behind `add` -- unmapped memory when the tensor ends a segment)
        for (int e = 0; e < nvalid; e++) {
          const int pe = (int)p + e;
          const int t = pe / hw;
          const int rem = pe - t * hw;
          const int h = rem / a.eW, w = rem - h * a.eW;
          if (((h | w) & 1) == 0) {
            const long long oa = ((((long long)n * a.M + m) * T_ + t) * Hh + (h >> 1)) * Wh + (w >> 1);
            val[e] += to_f<T>(((const T*)a.add)[oa]);
          }
        }
      } else if constexpr (EPI == X3D_EPI_SWISH_BWD) {
        float b[8];
        if constexpr (OVEC == 8) {
#pragma unroll
          for (int e = 0; e < 8; e++) b[e] = (float)epl8[i][e];
        } else {
          for (int e = 0; e < 8; e++) b[e] = (e < nvalid) ? to_f<T>(((const T*)a.braw)[o + e]) : 0.f;
        }
        const SwishCoef sc_ = swish_coef(esb[i], etb[i], egt[i]);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float xh_, d_;
          swish_bwd_(sc_, b[e], xh_, d_);
          const float dv = val[e] * d_;
          val[e] = dv;
          if (e < nvalid && rvalid) {   // sums of the fp32 values: equal to the sums of the stored (rounded) ones to ~2^-9/sqrt(count)
            st1[i] += dv;
            st2[i] += dv * b[e];
          }
        }
      }
      if constexpr (EPI == EPI_STATS) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          if (e < nvalid && rvalid) {
            st1[i] += val[e];
            st2[i] += val[e] * val[e];
          }
        }
      }
      if constexpr (BSTORE) {
        hx8 ov;
#pragma unroll
        for (int e = 0; e < 8; e++) ov[e] = (H)val[e];
        // byte offset inside sample n's [M][P] matrix; 0x80000000 is past num_records -> the store is discarded
        const unsigned off = (rvalid && (!RAG || nvalid == 8)) ? (unsigned)(((long long)m * a.P + p) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, ov), yrsrc, off, 0, 0);
        if constexpr (RAG) {
          if (rvalid && nvalid < 8) store8_ragged<T, hx8>((T*)a.y + o, ov, nvalid);
        }
      } else if constexpr (OVEC == 8) {
        if (!RAG || nvalid == 8) VecIO<T, 8>::store((T*)a.y + o, val);
        else {
          for (int e = 0; e < nvalid; e++) ((T*)a.y)[o + e] = from_f<T>(val[e]);
        }
      } else {
        for (int e = 0; e < nvalid; e++) ((T*)a.y)[o + e] = from_f<T>(val[e]);
      }
    }
    }
    // Os is rewritten only after the next tile's first two barriers
  }

  if (tile_begin < tile_end) flush_sums(n_prev);
}

static inline size_t pw_bf16_lds_bytes(int mt, int K) {
  const int slab = mt <= 2 ? mt : 1;
  const size_t xb = (size_t)PWB_KCH * PWB_XP * 2, ob = (size_t)slab * 32 * PWB_OP * 4;
  return (xb > ob ? xb : ob) + (size_t)mt * 32 * (((K + 15) & ~15) + 8) * 2 + (K < 320 ? (size_t)((K + 15) & ~15) * 16 : (size_t)((K + 15) & ~15) * 8);
}

// rows of 32 output channels per workgroup.  Every extra row block (gy > 1) re-stages the activation tile and
// repeats its prologue, but tall panels serialise the epilogue and cost registers; measured on X3D-M (r01c):
// three row tiles in one panel win when they remove the second row block (M in 65..96: 133 -> 91 us on the
// 216->96 layers), taller panels lose (96->216: 69 -> 118 us with MT = 7), and K >= 320 panels are too wide
// for more than 32 rows at three workgroups per CU.
static inline int pw_bf16_pick_mt(int M, int K) {
  const int mt = ceil_div(M, 32);
  // experiment hook: X3D_PW_MTMAP="7:4,14:7" maps a row-tile count to a panel height
#ifdef X3D_EXPERIMENTS
  const char* map = getenv("X3D_PW_MTMAP");
#else
  const char* map = nullptr;
#endif
  if (map) {
    for (const char* q = map; *q;) {
      const int key = atoi(q);
      const char* c = strchr(q, ':');
      if (!c) break;
      const int val = atoi(c + 1);
      if (key == mt && (val == 1 || val == 2 || val == 3 || val == 4 || val == 7) && pw_bf16_lds_bytes(val, K) <= 160 * 1024) return val;
      const char* n = strchr(c, ',');
      if (!n) break;
      q = n + 1;
    }
  }
  if (mt <= 1 || K >= 320) return 1;
  if (mt == 3 && pw_bf16_lds_bytes(3, K) <= 80 * 1024) return 3;
  return 2;
}

template <typename H, int VEC, int MT, int PRO, int EPI, int STRIDED, int OVEC, bool RAG = false>
static int pw_bf16_launch_cfg(PwGemmArgs& a, hipStream_t st) {
  if constexpr (!RAG && VEC == 8 && OVEC == 8) {
    if (a.P % 8 != 0) return pw_bf16_launch_cfg<H, VEC, MT, PRO, EPI, STRIDED, OVEC, true>(a, st);   // ragged rows
  }
  constexpr int BM = MT * 32, BN = PWB_BN;
  a.KC = (a.K + 15) & ~15;
  const size_t lds = pw_bf16_lds_bytes(MT, a.K);
  X3D_REQUIRE(lds <= 160 * 1024, "pw_gemm_bf16: K = %d needs %zu B of LDS", a.K, lds);
  const int gy = ceil_div(a.M, BM);
  const long long total_tiles = ceil_div_ll(a.P, BN) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_gemm_bf16: too many tiles");
  X3D_REQUIRE((long long)a.M * a.P * 2 < (1ll << 31), "pw_gemm_bf16: one sample's output exceeds the 2 GB buffer-store window");
  X3D_DESCRIBE("pw_gemm_bf16_kernel<%s, %d, %d, %d, %d, %d, %d, %d>", HV<H>::name, VEC, MT, PRO, EPI, STRIDED, OVEC, (int)RAG);
  auto kern = pw_gemm_bf16_kernel<H, VEC, MT, PRO, EPI, STRIDED, OVEC, RAG>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
  }
  // One balanced round: as many workgroups as the chip holds at once (occupancy of THIS instantiation at THIS
  // LDS size x CUs), the point tiles split evenly among them -- a fixed tile count per workgroup left partial
  // last rounds (e.g. 1568 workgroups on 768 slots).  The occupancy is cached per LDS size.
  static size_t occ_lds[8];
  static int occ_slots[8], occ_n = 0;
  int slots = 0;
  for (int i = 0; i < occ_n; i++) if (occ_lds[i] == lds) slots = occ_slots[i];
  if (slots == 0) {
    int nb = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    slots = nb * cus;
    if (occ_n < 8) { occ_lds[occ_n] = lds; occ_slots[occ_n] = slots; occ_n++; }
  }
  long long gx_target = slots / gy;
  if (gx_target < 1) gx_target = 1;
  int tpb = (int)ceil_div_ll(total_tiles, gx_target);
  // a wide weight panel (K*BM bf16 per workgroup, through L2) must be amortised over several tiles even if that
  // leaves fewer workgroups than slots (r01c sweep: 4 tiles is the optimum for K >= 192 with packed panels)
  int tpb_min = a.K >= 192 ? 4 : (a.K >= 96 ? 2 : 1);
  tpb_min = x3d_env_int("X3D_PW_TPBMIN", tpb_min);   // experiment hook
  if (tpb < tpb_min) tpb = tpb_min;
  a.tiles_per_block = tpb;
  const long long gx = ceil_div_ll(total_tiles, tpb);
  // widest aligned fp32 vector along the contiguous axis of the weight matrix
  {
    const int contig = (a.wsk == 1) ? a.K : a.M;          // row length of the source
    int wv = 4;
    while (wv > 1 && ((contig % wv) != 0 || ((uintptr_t)a.w % (wv * 4)) != 0)) wv >>= 1;
    if (a.wsk != 1 && (BM % wv) != 0) wv = 1;
    a.wvec = wv;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_gemm_bf16");
  return X3D_OK;
}

template <typename H, int VEC, int PRO, int EPI, int STRIDED, int OVEC>
static int pw_bf16_launch_tile(PwGemmArgs& a, hipStream_t st) {
  int mt = pw_bf16_pick_mt(a.M, a.K);
  if constexpr (VEC == 1) mt = mt > 2 ? 2 : mt;   // scalar fallback path: 32 staging registers per chunk, keep the panel small
  switch (mt) {
    case 7: if constexpr (VEC != 1) return pw_bf16_launch_cfg<H, VEC, 7, PRO, EPI, STRIDED, OVEC>(a, st);
    case 4: if constexpr (VEC != 1) return pw_bf16_launch_cfg<H, VEC, 4, PRO, EPI, STRIDED, OVEC>(a, st);
    case 3: if constexpr (VEC != 1) return pw_bf16_launch_cfg<H, VEC, 3, PRO, EPI, STRIDED, OVEC>(a, st);
    case 2: return pw_bf16_launch_cfg<H, VEC, 2, PRO, EPI, STRIDED, OVEC>(a, st);
    default: return pw_bf16_launch_cfg<H, VEC, 1, PRO, EPI, STRIDED, OVEC>(a, st);
  }
}

// vec: common alignment (elements) of the streamed inputs; ovec: of the outputs / epilogue tensors
template <typename H, int PRO, int EPI>
static int pw_bf16_launch_vec(PwGemmArgs& a, int vec, int ovec, hipStream_t st) {
  if (a.stride > 1) {
    if constexpr (PRO == PRO_NONE && EPI == EPI_STATS) {
      const int gv = (a.stride == 2 && vec >= 1 && ovec >= 8) ? strided_gather_gv(a.W, a.Wo, a.P, a.x) : 0;
      if (gv == 4) return pw_bf16_launch_tile<H, 8, PRO, EPI, 4, 8>(a, st);
      if (gv == 2) return pw_bf16_launch_tile<H, 8, PRO, EPI, 2, 8>(a, st);
      if (gv == 1) return pw_bf16_launch_tile<H, 8, PRO, EPI, 1, 8>(a, st);
      if (ovec >= 8 && a.P % 8 == 0) return pw_bf16_launch_tile<H, 1, PRO, EPI, 1, 8>(a, st);   // (whole output vectors only)
      return pw_bf16_launch_tile<H, 1, PRO, EPI, 1, 1>(a, st);
    } else {
      x3d_set_error("pw: strided gather only in forward");
      return X3D_ERR_INVALID;
    }
  }
  if (vec >= 8 && ovec >= 8) return pw_bf16_launch_tile<H, 8, PRO, EPI, 0, 8>(a, st);
  return pw_bf16_launch_tile<H, 1, PRO, EPI, 0, 1>(a, st);
}
