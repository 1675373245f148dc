// Weights-STATIONARY variant of the bf16 pointwise GEMM for the deep, narrow layers (X3D stage 5: 192 <-> 432 channels
// on 784 points per sample, 50k points per launch at batch 64).
//
// The weights-streamed kernel (pw_gemm_ws.h) re-reads its A operand (the weights) from L2 for every 32-point tile; at
// two workgroups per CU every k-step group, every epilogue load and every staging round exposes its full latency (SQ
// counters, r01n: instructions active 18-27 % of the wave cycles) and a 62 MB layer takes 52 us.  Here ONE persistent
// workgroup per CU keeps the weights in REGISTERS for its whole life: wave w owns row blocks w, w + NW, ... (RB of them,
// KS k-steps x 4 VGPRs each: 108 VGPRs for 432 -> 192, 96 for 192 -> 432), loaded once from the MFMA-operand-tiled
// panel image (pw_pack.hip: one contiguous 1 KB wave load per k-step).  Per 32-point tile only the streamed operand
// moves: global -> registers one tile ahead -> (prologue) -> one of TWO LDS tiles, so that the commit of tile t+1
// follows the MFMAs of tile t with a single barrier per tile; the MFMA loop is KS fully unrolled steps of
// ds_read_b64_tr_b16 + RB MFMAs with no global access in it.  Epilogue operands (residual / swish' input) are
// fetched before the MFMA loop.
#pragma once
#include "pw_gemm_ws.h"

// waves per workgroup: all of them stage (two per SIMD: the swish prologue is transcendental-bound VALU work and wants
// the four SIMDs evenly loaded), the first NW of them own row blocks
#define WST_NWT 8

// RAG (round 2): rows of P % 8 != 0 points (16-bit X3D-S / XS stage 5).  A row's last vector is loaded from
// row_end - 8 -- always inside the tensor, still one unconditional load per slot -- and moved into place when it is
// consumed (shift_down8, common.h); its stores go element by element and its sums are masked.
// NWT: waves per workgroup.  8 everywhere but for the one shape whose stationary operand does not fit the 256 registers a
// wave has at two waves per SIMD (K = 630: 40 k-steps x 4 VGPRs = 160 + staging): that one runs four waves (one per SIMD,
// 512 registers each: the weights live in the accumulation-register half of the file).
template <typename H, int PRO, int EPI, int NW, int RB, int KS, int OCC, bool RAG = false, int NWT = WST_NWT>
__global__ __launch_bounds__(NWT * 64, OCC) void pw_gemm_wst_kernel(const PwGemmArgs a) {
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4; typedef typename HV<H>::x2 hx2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BN = WS_BN, OP = WS_OP, NT = NWT * 64, Kp = KS * 16, WP = Kp + 8;
  static_assert(NW <= NWT, "MFMA waves are a subset of the workgroup");
  constexpr bool HAS_SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  constexpr bool BNA_ = (EPI == EPI_BNADD);
  constexpr bool EPI_LOADS = (EPI == X3D_EPI_ADD) || (EPI == X3D_EPI_SWISH_BWD) || BNA_;
  constexpr int CSW = (PRO == PRO_AFFINE) ? 2 : 4;
  constexpr int NSV = (Kp * 4 + NT - 1) / NT;             // staging vectors (8 points) per thread
  H* Xs = (H*)smem_raw;                                               // [2][Kp][32]
  float* Cs = (float*)(smem_raw + (size_t)2 * Kp * 64);                     // [Kp][CSW]
  float* Os = (float*)(smem_raw + (size_t)2 * Kp * 64 + (size_t)Kp * 16);   // [NW waves][32][OP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int mt = (a.M + 31) >> 5;
  // wide outputs (X3D-XL: M = 280 / 306 / 630 with K up to 630): the row blocks are split over blockIdx.y -- each slice is
  // its own persistent workgroup set with NW * RB stationary row blocks (the streamed operand is staged once per slice)
  const int mib = blockIdx.y * (NW * RB);
  const int rsh = 8 - (int)(a.P & 7);   // RAG: places a row's last vector (loaded from row_end - 8) moves down by
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  if (tile_begin >= tile_end) return;
  float* myOs = Os + wid * 32 * OP;

  // ---- the stationary operand
  hx8 A[RB][KS];
#pragma unroll
  for (int j = 0; j < RB; j++) {
    const int mi = mib + wid + NW * j;
    if (wid < NW && mi < mt) {
      const H* wt = (const H*)a.wp + (long long)a.wp_rows * WP + ((long long)mi * KS * 64 + lane) * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) A[j][ks] = *(const hx8*)(wt + ks * 512);
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) A[j][ks][e] = (H)0.f;
    }
  }

  auto fill_coef = [&](int n) {
    if constexpr (PRO != PRO_NONE) {
      for (int k = tid; k < Kp; k += NT) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        if (k < a.K) {
          if constexpr (PRO == PRO_AFFINE) {
            const float g = a.gate ? a.gate[(long long)n * a.K + k] : 1.0f;
            c[0] = a.coef[k * 2] * g;
            c[1] = a.coef[k * 2 + 1] * g;
          } else if constexpr (PRO == PRO_TAIL) {   // s_c*x + (s_r|1)*x2 + (t_c + t_r|0), as in pw_gemm_bf16.h
            c[0] = a.coef[k * 2]; c[1] = a.coef2 ? a.coef2[k * 2] : 1.0f;
            c[2] = a.coef[k * 2 + 1] + (a.coef2 ? a.coef2[k * 2 + 1] : 0.f);
          } else {   // (filled once per workgroup: the publishing thread of channel k is unique in the launch)
            float cA_, cB_, cC_;
            bn_bwd_coef_load(a.coef, a.fold, k, blockIdx.x == 0 && blockIdx.y == 0, cA_, cB_, cC_);
            c[0] = cA_; c[1] = cB_; c[2] = cC_;
          }
        }
        if constexpr (CSW == 2) *(float2*)&Cs[k * 2] = make_float2(c[0], c[1]);
        else *(f32x4*)&Cs[k * 4] = c;
      }
    }
  };

  // ---- staging: vector v = tid + NT*i -> row v >> 2, 8 points at unit v & 3.  Loads are unconditional (clamped
  // address, zero selected afterwards) so that the whole batch is in flight at once.
  // TWO register sets: the loads of tile t+2 are issued before the commit of tile t+1, so a whole iteration (commit,
  // epilogue, barrier, MFMAs) covers their latency.  Tiles past the end re-load the last tile (unconditional: the
  // number of loads in flight stays static and the compiler can wait with an exact vmcnt).
  constexpr bool TWO = (PRO == PRO_BNBWD) || (PRO == PRO_TAIL);   // a second streamed tensor (a.x2)
  constexpr int NSY = TWO ? NSV : 1;
  hx8 xr0[NSV], yr0[NSY], xr1[NSV], yr1[NSY];
  auto issue_loads = [&](int tile_, hx8 (&xr)[NSV], hx8 (&yr)[NSY]) {
    const int tile = a.hot ? tile_begin : min(tile_, tile_end - 1);
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * NT;
      const int k = v >> 2;
      const long long p = p0 + (v & 3) * 8;
      const bool ok = k < a.K && p < a.P;
      const long long pl = (RAG && a.P - p < 8) ? a.P - 8 : p;      // the row's last, partial vector: from row_end - 8
      const long long o = ok ? ((long long)n * a.K + k) * a.Pin + pl : 0;
      xr[i] = *(const hx8*)((const T*)a.x + o);
      if constexpr (TWO) yr[i] = *(const hx8*)((const T*)a.x2 + o);
    }
  };
  auto commit = [&](int tile, H* dstbuf, const hx8 (&xr)[NSV], const hx8 (&yr)[NSY]) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * NT;
      const int k = v >> 2;
      if (k >= Kp) continue;
      const bool ok = k < a.K && p0 + (v & 3) * 8 < a.P;
      H* dst = &dstbuf[k * BN + (v & 3) * 8];
      hx8 z;
#pragma unroll
      for (int e = 0; e < 8; e++) z[e] = (H)0.f;
      hx8 xv = xr[i], yv = yr[TWO ? i : 0];
      if constexpr (RAG) {
        if (ok && a.P - (p0 + (v & 3) * 8) < 8) {
          xv = shift_down8(xv, rsh);
          if constexpr (TWO) yv = shift_down8(yv, rsh);
        }
      }
      if constexpr (PRO == PRO_NONE) {
        *(hx8*)dst = ok ? xv : z;
      } else {
        float val[8];
#pragma unroll
        for (int e = 0; e < 8; e++) val[e] = (float)xv[e];
        if constexpr (PRO == PRO_AFFINE) {
          const float2 cf = *(const float2*)&Cs[k * 2];
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] = cf.x * val[e] + cf.y;
          act_vec<8>(val, a.act);
        } else {
          const f32x4 cf = *(const f32x4*)&Cs[k * 4];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            val[e] = cf[0] * val[e] + cf[1] * (float)yv[e] + cf[2];
            if constexpr (PRO == PRO_TAIL) val[e] = fmaxf(val[e], 0.f);
          }
        }
        if (!ok) {
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] = 0.f;
        }
        VecIO<H, 8>::store(dst, val);
        if constexpr (PRO == PRO_TAIL) {
          // the activated input IS the output y of the block below: kept for its other readers (the next shortcut, the backward
          // pass) -- what the separate x3d_tail_fwd pass would have written
          if (ok && blockIdx.y == 0) {
            const long long p_ = p0 + (v & 3) * 8;
            T* yd = (T*)a.ystore + ((long long)n * a.K + k) * a.Pin + p_;
            if (!RAG || a.P - p_ >= 8) VecIO<T, 8>::store(yd, val);
            else for (int e = 0; e < (int)(a.P - p_); e++) yd[e] = from_f<T>(val[e]);
          }
        }
      }
    }
  };

  // per-lane partial sums: lane owns row (lane >> 1) of each of this wave's row blocks, 16 points
  float st1[HAS_SUMS ? RB : 1], st2[HAS_SUMS ? RB : 1];
  if constexpr (HAS_SUMS) {
#pragma unroll
    for (int i = 0; i < RB; i++) { st1[i] = 0.f; st2[i] = 0.f; }
  }
  auto flush_sums = [&](int n) {
    if constexpr (HAS_SUMS) {
#pragma unroll
      for (int i = 0; i < RB; i++) {
        const int mi = mib + wid + NW * i;
        const float s1 = st1[i] + dpp_get<0xB1, 0xF>(st1[i]), s2 = st2[i] + dpp_get<0xB1, 0xF>(st2[i]);
        const int m = mi * 32 + (lane >> 1);
        if (wid < NW && mi < mt && (lane & 1) == 0 && m < a.M) {
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

  const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_off = (8 * (g16 >> 1) + q) * BN + 16 * (g16 & 1) + 4 * pp;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const int row = lane >> 1, c0 = 16 * (lane & 1);

  // EPI_BNADD: this lane's output rows and their coefficients are the same for every tile -- loaded once (inside the tile
  // loop the loads sit behind the staging prefetches and their wait is a vmcnt(0): a full memory latency per tile)
  float bna_s[BNA_ ? RB : 1], bna_t[BNA_ ? RB : 1], bna_g[BNA_ ? RB : 1];
  if constexpr (BNA_) {
#pragma unroll
    for (int j = 0; j < RB; j++) {
      const int m = (mib + wid + NW * j) * 32 + row;
      bnadd_coef(a, m, wid < NW && m < a.M, bna_s[j], bna_t[j], bna_g[j]);
      if (!a.add) bna_g[j] = 0.f;
    }
  }

  float swb_s[EPI == X3D_EPI_SWISH_BWD ? RB : 1], swb_t[EPI == X3D_EPI_SWISH_BWD ? RB : 1];   // BN_b scale / shift of this lane's rows
  if constexpr (EPI == X3D_EPI_SWISH_BWD) {
#pragma unroll
    for (int j = 0; j < RB; j++) {
      const int m = (mib + wid + NW * j) * 32 + row;
      const bool okm = wid < NW && m < a.M;
      swb_s[j] = okm ? a.b_ss[m * 2] : 0.f;
      swb_t[j] = okm ? a.b_ss[m * 2 + 1] : 0.f;
    }
  }

  int n_prev = tile_begin / tiles_per_n;
  fill_coef(n_prev);
  issue_loads(tile_begin, xr0, yr0);
  __syncthreads();
  commit(tile_begin, Xs, xr0, yr0);
  __syncthreads();
  issue_loads(tile_begin + 1, xr1, yr1);
  issue_loads(tile_begin + 2, xr0, yr0);

  // one tile: MFMAs of `tile`, commit of tile + 1 out of the register set (xr, yr), re-load that set with tile + 3
  auto step = [&](int tile, int cur, hx8 (&xr)[NSV], hx8 (&yr)[NSY]) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    if (n != n_prev) {
      if constexpr (EPI == X3D_EPI_SWISH_BWD) flush_sums(n_prev);
    }
    n_prev = n;

    // ---- epilogue operands of this tile, in flight during the MFMAs.  The SE gate of this tile's sample goes FIRST: loaded
    // next to its use in the epilogue it was the youngest load in flight, and waiting for it a vmcnt(0) -- every staging
    // prefetch issued before it had to land too (in-order retirement), once per tile
    float swg[EPI == X3D_EPI_SWISH_BWD ? RB : 1];
    if constexpr (EPI == X3D_EPI_SWISH_BWD) {
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const int m = (mib + wid + NW * j) * 32 + row;
        const bool okg = a.egate && wid < NW && m < a.M;
        const float gl = (a.egate ? a.egate : a.b_ss)[okg ? (long long)n * a.M + m : 0];   // unconditional, clamped
        swg[j] = okg ? gl : 1.0f;
      }
    }
    hx8 eo[EPI_LOADS ? RB : 1][2];
    f32x16 acc[RB];
    H es[EPI == X3D_EPI_ADD_STRIDED ? RB : 1][2][4];      // strided shortcut gradient: one value per even pixel
    if (wid < NW) {
    if constexpr (EPI == X3D_EPI_ADD_STRIDED) {
      // dx [eH x eW] receives `add` [ceil(eH/2) x ceil(eW/2)] on its even pixels.  eW is even (dispatch): the 8 points of
      // a vector are four pairs inside one image row each, the first of a pair on an even column.  32-bit index maths.
      const int hw = a.eH * a.eW, Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1, T_ = (int)a.P / hw;
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const int m = (mib + wid + NW * j) * 32 + row;
        const T* abase = (const T*)a.add + ((long long)n * a.M + min(m, a.M - 1)) * T_ * Hh * Wh;
#pragma unroll
        for (int hv = 0; hv < 2; hv++)
#pragma unroll
          for (int gq = 0; gq < 4; gq++) {
            const int pe = (int)p0 + c0 + 8 * hv + 2 * gq;
            const int t = pe / hw, rem = pe - t * hw;
            const int h = rem / a.eW, w = rem - h * a.eW;
            const bool ok = m < a.M && pe < (int)a.P && (h & 1) == 0;
            const H v = abase[ok ? (t * Hh + (h >> 1)) * Wh + (w >> 1) : 0];
            es[j][hv][gq] = ok ? v : (H)0.f;
          }
      }
    }
    if constexpr (EPI_LOADS) {
      // (BNADD without a residual: the loads go to the output tensor's own first vector -- in range, value unused)
      const T* src = (const T*)(EPI == X3D_EPI_SWISH_BWD ? a.braw : (BNA_ && !a.add) ? a.y : a.add);
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const int m = (mib + wid + NW * j) * 32 + row;
#pragma unroll
        for (int hv = 0; hv < 2; hv++) {
          const long long p = p0 + c0 + 8 * hv;
          const long long pl = (RAG && a.P - p < 8) ? a.P - 8 : p;
          const long long o = (m < a.M && p < a.P && !(BNA_ && !a.add)) ? ((long long)n * a.M + m) * a.P + pl : 0;
          eo[j][hv] = *(const hx8*)(src + o);
        }
      }
    }

    // ---- MFMAs: B operand from the current LDS tile, A from registers
    const H* xb = Xs + cur * (Kp * BN) + tr_off;
#pragma unroll
    for (int j = 0; j < RB; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb + ks * 16 * BN));
      const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb + (ks * 16 + 4) * BN));
      const s16x8 bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      const hx8 bv = __builtin_bit_cast(hx8, bs);
#pragma unroll
      for (int j = 0; j < RB; j++) acc[j] = mfma16<H>(A[j][ks], bv, acc[j]);
    }

    }   // wid < NW

    // ---- the next tile goes into the other LDS tile.  This sits BEFORE the epilogue: the commit waits for the staged
    // loads with vmcnt, which retires in order and counts stores -- after the epilogue it would also wait for this
    // tile's output stores, every tile
    if (tile + 1 < tile_end) {
      if constexpr (PRO == PRO_AFFINE) {
        const int n1 = (tile + 1) / tiles_per_n;
        if (a.gate && n1 != n) {        // every earlier read of the table is behind the previous barrier
          fill_coef(n1);
          __syncthreads();
        }
      }
      commit(tile + 1, Xs + (cur ^ 1) * (Kp * BN), xr, yr);
    }

    if (wid < NW) {
    // ---- epilogue through the wave-private slab: lane -> row lane >> 1, points 16*(lane & 1) .. +15
#pragma unroll
    for (int j = 0; j < RB; j++) {
      const int mi = mib + wid + NW * j;
      if (mi >= mt) continue;
#pragma unroll
      for (int e = 0; e < 16; e++) myOs[((e & 3) + 8 * (e >> 2) + 4 * half) * OP + r] = acc[j][e];
      const int m = mi * 32 + row;
      if (m < a.M) {
        float sb = 0.f, tb = 0.f, gt = 1.f;
        if constexpr (EPI == X3D_EPI_SWISH_BWD) { sb = swb_s[j]; tb = swb_t[j]; gt = swg[j]; }
        if constexpr (BNA_) { sb = bna_s[j]; tb = bna_t[j]; gt = bna_g[j]; }
#pragma unroll
        for (int hv = 0; hv < 2; hv++) {
          const long long p = p0 + c0 + 8 * hv;
          if (p >= a.P) continue;                       // P % 8 == 0: a vector of 8 points is inside or outside
          const long long o = ((long long)n * a.M + m) * a.P + p;
          const int left = RAG ? (int)min((long long)8, a.P - p) : 8;   // RAG: the row may end inside this vector
          if constexpr (RAG && EPI_LOADS) {
            if (left < 8) eo[j][hv] = shift_down8(eo[j][hv], rsh);
          }
          float val[8];
          {
            const f32x4 v0 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv], v1 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv + 4];
#pragma unroll
            for (int e = 0; e < 4; e++) { val[e] = v0[e]; val[4 + e] = v1[e]; }
          }
          if constexpr (EPI == X3D_EPI_ADD) {
#pragma unroll
            for (int e = 0; e < 8; e++) val[e] += (float)eo[j][hv][e];
          } else if constexpr (BNA_) {
            const float lo = a.eact == X3D_ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
            for (int e = 0; e < 8; e++) val[e] = fmaxf(sb * val[e] + tb + (gt != 0.f ? gt * (float)eo[j][hv][e] : 0.f), lo);
          } else if constexpr (EPI == X3D_EPI_ADD_STRIDED) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) val[2 * gq] += (float)es[j][hv][gq];
          } else if constexpr (EPI == X3D_EPI_SWISH_BWD) {
            const SwishCoef sc_ = swish_coef(sb, tb, gt);
#pragma unroll
            for (int e = 0; e < 8; e++) {
              const float b = (float)eo[j][hv][e];
              float xh_, d_;
              swish_bwd_(sc_, b, xh_, d_);
              const float dv = val[e] * d_;
              val[e] = dv;
              if (!RAG || e < left) {
                st1[j] += dv;
                st2[j] += dv * b;
              }
            }
          }
          if constexpr (EPI == EPI_STATS) {
#pragma unroll
            for (int e = 0; e < 8; e++) if (!RAG || e < left) { st1[j] += val[e]; st2[j] += val[e] * val[e]; }
          }
          if (!RAG || left == 8) VecIO<T, 8>::store((T*)a.y + o, val);
          else {
            for (int e = 0; e < left; e++) ((T*)a.y)[o + e] = from_f<T>(val[e]);
          }
        }
      }
    }

    }   // wid < NW
    __syncthreads();                    // one barrier per tile
    issue_loads(tile + 3, xr, yr);
  };
  for (int tile = tile_begin; tile < tile_end; tile += 2) {
    step(tile, 0, xr1, yr1);
    if (tile + 1 < tile_end) step(tile + 1, 1, xr0, yr0);
  }
  flush_sums(n_prev);
}

template <int NW, int KS>
static inline size_t pw_wst_lds_bytes() {
  return (size_t)2 * KS * 16 * 64 + (size_t)KS * 16 * 16 + (size_t)NW * 32 * WS_OP * 4;
}

// shapes with an instantiation: 0 = none, 1 = K 417..432 -> M <= 192 (6 waves x 1 row block x 27 k-steps),
// 2 = K 177..192 -> M <= 448 (7 waves x 2 row blocks x 12 k-steps); stage 4, two workgroups per CU (<= 128 VGPRs: one
// workgroup's prologue overlaps the other's MFMAs): 3 = K 209..224 -> M <= 96 (3 x 1 x 14), 4 = K 81..96 -> M <= 224
// (7 x 1 x 6), 5 = K 81..96 -> M <= 448 (7 x 2 x 6)
static inline int pw_wst_shape(const PwGemmArgs& a, int vec, int ovec) {
  const int e_wst = x3d_env_int("X3D_PW_WST", -1);   // A/B switch: 0 = never, 1 = stage 5 only
  if (e_wst == 0) return 0;
  if (!a.wp || vec < 8 || ovec < 8 || a.stride != 1 || a.P < 8) return 0;   // (P % 8 != 0: the RAG instantiations)
  const int ks = (a.K + 15) >> 4;
  if (ks == 27 && a.M <= 192) return 1;
  if (ks == 12 && a.M <= 448) return 2;
  if (e_wst == 1) return 0;
  if (ks == 14 && a.M <= 96) return 3;
  if (ks == 6 && a.M <= 224) return 4;
  if (ks == 6 && a.M <= 448) return 5;     // stage-5 block 0 `a` conv: 96 -> 432 (7 x 2 x 6)
  // X3D-XL widths (72 / 162, 136 / 306, 280 / 630; round 3).  The resident-panel kernel re-reads every weight fragment from
  // LDS in every wave (>= 1 KB of LDS traffic per MFMA: LDS-bound below a third of the matrix-core rate) and repeats the
  // swish prologue per 32- / 64- / 96-row block; measured on 60 clips of 16x312x312 in fp16: 630 -> 280 279 us (floor 22),
  // 306 -> 136 237 us (floor 42), 280 -> 630 108 us, 136 -> 306 137 us.  Row blocks beyond NW * RB go to blockIdx.y.
  if (x3d_env_int("X3D_PW_WST_XL", 1) == 0) return 0;   // A/B switch: 0 = off
  if (ks == 20 && a.M <= 160) return 6;    // stage-4 `c`: 306 -> 136 (5 waves x 1 row block x 20 k-steps)
  if (ks == 9 && a.M <= 640) return 7;     // stage-4 `a`: 136 -> 306 (5 x 2 x 9); stage-5 block 0: 136 -> 630 in two slices
  // (round 4: the first slice storing the activated tile in place and the other two running without the prologue as a second
  // launch: 168 vs 166 us -- a slice takes its 55 us with or without the prologue; removed again)
  if (ks == 40 && a.M <= 288) return 8;    // stage-5 `c`: 630 -> 280 (3 x 1 x 40, three slices)
  if (ks == 18 && a.M <= 640) return 9;    // stage-5 `a` / conv5: 280 -> 630 (5 x 2 x 18, two slices)
  if (ks == 11 && a.M <= 96) return 10;    // stage-3 `c`: 162 -> 72 (3 x 1 x 11)
  // (72 -> 162, one slice, measured 227 us against 209 on the resident-panel kernel: only the two-slice case runs here)
  if (ks == 5 && a.M > 192 && a.M <= 384) return 11;    // stage-4 block 0 `a`: 72 -> 306 (6 x 1 x 5, two slices)
  return 0;
}
// shapes 6.. exist for one kind of prologue only (the `c` convs carry BN_b * gate -> swish, the `a` convs none): the other
// combination has no instantiation and falls back to the resident-panel kernel
static inline bool pw_wst_shape_has_prologue(int shape) { return shape == 6 || shape == 8 || shape == 10; }

template <typename H, int PRO, int EPI, int NW, int RB, int KS, int OCC, bool RAG = false, int NWT = WST_NWT>
static int pw_wst_launch_t(PwGemmArgs& a, hipStream_t st) {
  if constexpr (!RAG) {
    if (a.P % 8 != 0) return pw_wst_launch_t<H, PRO, EPI, NW, RB, KS, OCC, true, NWT>(a, st);
  }
  a.KC = KS * 16;
  const size_t lds = pw_wst_lds_bytes<NW, KS>();
  X3D_DESCRIBE("pw_gemm_wst_kernel<%s, %d, %d, %d, %d, %d, %d, %d>", HV<H>::name, PRO, EPI, NW, RB, KS, OCC, (int)RAG);
  auto kern = pw_gemm_wst_kernel<H, PRO, EPI, NW, RB, KS, OCC, RAG, NWT>;
  static bool attr_set = false;
  static int cus = 256;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    attr_set = true;
  }
  const long long total_tiles = ceil_div_ll(a.P, WS_BN) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_gemm_wst: too many tiles");
  const int gy = ceil_div(ceil_div(a.M, 32), NW * RB);       // row-block slices (1 for every X3D-S / M / L layer)
  const long long slots = (long long)cus * OCC / gy > 0 ? (long long)cus * OCC / gy : 1;
  const long long tpb = ceil_div_ll(total_tiles, slots);
  a.tiles_per_block = (int)tpb;
  a.hot = x3d_env_int("X3D_PW_WST_HOT", 0);
  const long long gx = ceil_div_ll(total_tiles, tpb);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(NWT * 64), lds, st, a);
  X3D_LAUNCH_CHECK("pw_gemm_wst");
  return X3D_OK;
}

// the folded residual tail (PRO_TAIL, training epilogue) on the stage-4 / stage-5 `a` convs and conv5 -- the shapes those layers take
static inline bool pw_wst_shape_has_tail(int shape) { return shape == 2 || shape == 4 || shape == 5; }
template <typename H>
static int pw_wst_launch_tail(PwGemmArgs& a, int shape, hipStream_t st) {
  switch (shape) {
    case 2: return pw_wst_launch_t<H, PRO_TAIL, EPI_STATS, 7, 2, 12, 1>(a, st);
    case 4: return pw_wst_launch_t<H, PRO_TAIL, EPI_STATS, 7, 1, 6, 2>(a, st);
    case 5: return pw_wst_launch_t<H, PRO_TAIL, EPI_STATS, 7, 2, 6, 2>(a, st);
  }
  x3d_set_error("pw_gemm_wst: shape %d has no tail instantiation", shape);
  return X3D_ERR_INVALID;
}

template <typename H, int PRO, int EPI>
static int pw_wst_launch(PwGemmArgs& a, int shape, hipStream_t st) {
  switch (shape) {
    // (stage 5 at two workgroups per CU over two row slices -- <3, 1, 27, 2> with four waves, <7, 1, 12, 2> -- measured in round 4:
    // forward 31 -> 48 us and 26.5 -> 29 us, dgrad 53 -> 178 / 58 us: one workgroup per CU stays)
    case 1: return pw_wst_launch_t<H, PRO, EPI, 6, 1, 27, 1>(a, st);
    case 2: return pw_wst_launch_t<H, PRO, EPI, 7, 2, 12, 1>(a, st);
    case 3: return pw_wst_launch_t<H, PRO, EPI, 3, 1, 14, (PRO == PRO_BNBWD ? 1 : 2)>(a, st);   // BNBWD: 132-146 VGPRs (unused: pw_dgrad.hip)
    case 4: return pw_wst_launch_t<H, PRO, EPI, 7, 1, 6, 2>(a, st);
    case 5: return pw_wst_launch_t<H, PRO, EPI, 7, 2, 6, 2>(a, st);
  }
  // X3D-XL shapes: forward only, one prologue kind each (pw_wst_shape_has_prologue)
  if constexpr (PRO == PRO_AFFINE) {
    switch (shape) {
      case 6: return pw_wst_launch_t<H, PRO, EPI, 5, 1, 20, 1>(a, st);
      case 8: return pw_wst_launch_t<H, PRO, EPI, 3, 1, 40, 1, false, 4>(a, st);
      // (162 -> 72 as four 4-wave workgroups per CU, <3, 1, 11, 4, false, 4>: config 5 33.0 -> 33.3 ms, round 4 -- not kept)
      case 10: return pw_wst_launch_t<H, PRO, EPI, 3, 1, 11, 2>(a, st);
    }
  }
  if constexpr (PRO == PRO_NONE) {
    switch (shape) {
      case 7: return pw_wst_launch_t<H, PRO, EPI, 5, 2, 9, 2>(a, st);
      case 9: return pw_wst_launch_t<H, PRO, EPI, 5, 2, 18, 1>(a, st);
      case 11: return pw_wst_launch_t<H, PRO, EPI, 6, 1, 5, 2>(a, st);
    }
  }
  x3d_set_error("pw_gemm_wst: shape %d has no instantiation for this prologue", shape);
  return X3D_ERR_INVALID;
}
