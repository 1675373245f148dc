// x3d_pw_bwd: data AND weight gradient of a pointwise convolution in ONE pass over dY (bf16 storage).
//
// The separate kernels (pw_dgrad.hip, pw_wgrad.hip) each stream g and yraw to rebuild dY = A*g + B*yraw + C,
// and the `c` conv additionally streams the raw depthwise output twice (swish' in the dgrad epilogue, swish in
// the wgrad prologue).  On the stage-2/3 layers (<= 128 channels either side) those re-reads are ~40 % of the
// backward traffic of the layer.  Here a workgroup stages the dY tile [Co][128 points] once in LDS and uses it
//   * as the B operand of   dX[ci][p]  = sum_co W[co][ci] * dY[co][p]      (read transposed, ds_read_b64_tr_b16)
//   * as the A operand of   dW[co][ci] += sum_p dY[co][p] * Xh[ci][p]      (read row-wise, ds_read_b128)
// Xh is the conv input: for the `a` conv the block input (loaded with the tile), for the `c` conv
// swish(gate * bn_b(braw)) -- produced by the SWISH_BWD epilogue of the dX tile from the braw values it loads
// anyway (sigmoid shared between swish and swish').
//
// LDS pitch of the dY tile: the transposed read wants pitch = 64 mod 256 bytes, the row read wants an odd
// number of 16-byte units; both hold with pitch 320 B and the 16-byte units of row k XOR-swizzled by (k>>2)&3
// (permutes the four units of each 64-byte segment: the transposed read of a half-wave still covers one whole
// segment per row, the 16 rows of a b128 group land on 16 distinct slots).
//
// dW partials stay in accumulators across the tiles of a workgroup and are added to dw with fp32 atomics once.
#include <stdlib.h>

#include <type_traits>

#include "pw_gemm.h"

typedef __attribute__((ext_vector_type(4))) short s16x4_f;
typedef __attribute__((ext_vector_type(8))) short s16x8_f;

struct PwBwdArgs {
  const void* g; const void* yraw; const float* coef;   // dY = A*g + B*yraw + C   rows = Co
  const void* wp; int wp_rows;                          // dgrad panel [roundup(Ci,32)][Kp + 8] bf16
  void* dx;                                             // [N][Ci][P]
  const void* add; const void* braw; const float* b_ss; const float* egate; double* nc_sums;
  int eH, eW;
  const void* x;                                        // EPI != SWISH_BWD: conv input [N][Ci][P]
  float* dw;                                            // [Co][Ci]
  int N, Co, Ci, Kp;
  long long P;
  int tiles_per_block;
  // TAIL (ADD epilogues): dx is the gradient wrt the OUTPUT y of the previous residual block, whose Add + ReLU backward
  // (x3d_tail_bwd) is applied here: dx = [y > 0] * (W^T dY + add) with y = this conv's input x (already staged for dW),
  // tail_sums_c [Ci][2] += (sum dx, sum dx * tail_c), tail_sums_r likewise with tail_r (NULL: identity shortcut)
  const void* tail_c; const void* tail_r; double* tail_sums_c; double* tail_sums_r;
};

#define FB_BN 128
#define FB_YP 160    // dY pitch (elements): 320 B
#define FB_XP 136    // Xh pitch (elements): 272 B = 17 units
#define FB_OP 132    // fp32 output slab pitch

// waves per SIMD the register allocator must leave room for: the small panels (one or two dW tiles) fit three workgroups
// per CU in LDS; their kernels are full of barriers and latency (7 per tile), a third workgroup fills them
#ifndef FB_WAVES
#define FB_WAVES(MT, KT) (((MT) * (KT) <= 2) ? 3 : 2)
#endif
// MT: 32-row tiles of Ci (dX rows / dW columns); KT: 32-row tiles of Co (dY rows / dW rows)
// (with the tail epilogue the two-tile panels need ~190 VGPRs: two workgroups per CU instead of three beat 44-88 bytes of
// scratch traffic per lane in the tile loop -- 24<->54 @56x56: 273 us plain, 355 us with the spilling tail, r03e)
// E4V (strided-add epilogue): rows of 8k points -- the shortcut gradient of an output vector is one 8-byte load; false: element
// loads inside the epilogue.  A template parameter, not a runtime branch: the element path's conditional loads inside the tile
// loop made every wait of the iteration a vmcnt(0), whichever path ran.
template <typename H, int MT, int KT, int EPI, int TAIL = 0, bool E4V = true>
__global__ __launch_bounds__(256, (TAIL && MT * KT == 2) ? 2 : FB_WAVES(MT, KT)) void pw_bwd_fused_kernel(const PwBwdArgs a) {
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4; typedef typename HV<H>::x2 hx2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BN = FB_BN, YP = FB_YP, XP = FB_XP, OP = FB_OP;
  constexpr bool SWB = (EPI == X3D_EPI_SWISH_BWD);
  // TAIL: 0 = off, 1 = folded residual-tail backward (identity shortcut in the block the gradient leaves), 2 = ... with a
  // shortcut conv there (a second product sum).  Separate instantiations: the second operand costs 8-16 VGPRs
  static_assert(!(TAIL && SWB), "the tail backward belongs to the `a` conv (ADD epilogues)");
  constexpr bool TAILR = TAIL == 2;
  constexpr int NT = MT * KT;                       // dW tiles
  constexpr int TPW = (NT + 3) / 4;                 // dW tiles per wave
  constexpr int NKS = NT >= 4 ? 1 : 4 / NT;         // k-parts (points) per tile when there are fewer tiles than waves
  constexpr int NVY = KT * 2;                       // dY staging vectors per thread (each of g, yraw)
  constexpr int NVX = MT * 2;                       // x staging vectors per thread
  constexpr int ROWS_PT = MT * 2;                   // dX rows per thread over all slabs
  const int Kp = a.Kp, WP = Kp + 8;
  // LDS: dY tile | Xh tile | W panel | fp32 slab (the slab aliases the dY tile when the weight-gradient MFMAs
  // run before the epilogue, i.e. whenever Xh does not come out of the epilogue)
  // the widest `c` layers pass the slab in two 16-row halves: 8.4 KB less LDS keeps two workgroups per CU
  constexpr bool HALF_SLAB = SWB && MT == 4;
  constexpr size_t YS_B = (size_t)KT * 32 * YP * 2, XS_B = (size_t)MT * 32 * XP * 2, OS_B = (size_t)(HALF_SLAB ? 16 : 32) * OP * 4;
  H* Ys = (H*)smem_raw;
  H* Xs = (H*)(smem_raw + YS_B);
  H* Ws = (H*)(smem_raw + YS_B + XS_B);
  float* Os = SWB ? (float*)(smem_raw + YS_B + XS_B + (size_t)MT * 32 * WP * 2) : (float*)smem_raw;
  // BN-backward coefficients {A, B, C, 0} per dY row in LDS (zeros for padded rows), read once per tile and row
  float* Cs = (float*)(smem_raw + YS_B + XS_B + (size_t)MT * 32 * WP * 2 + (SWB ? OS_B : 0));
  static_assert(SWB || OS_B <= YS_B + XS_B, "slab must fit the dY + Xh tiles it aliases (both are rewritten by every commit)");

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  // blockIdx.y selects a slice of MT*32 input channels (dX rows / dW columns) when Ci is wider than one panel:
  // the dY tile is then staged once per slice (as x3d_pw_dgrad does per row block) but still feeds both products
  const int m0 = blockIdx.y * MT * 32;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);

  // ---- one-time LDS set-up: zero the dY / Xh tiles (padding rows stay zero), copy the weight panel
  {
    hx8 zero;
#pragma unroll
    for (int e = 0; e < 8; e++) zero[e] = (H)0.f;
    for (int i = tid; i < (int)((YS_B + XS_B) / 16); i += 256) ((hx8*)smem_raw)[i] = zero;
    const hx8* src = (const hx8*)((const H*)a.wp + (long long)m0 * WP);
    const int nvec = MT * 32 * WP / 8;
    const int lim = max(0, min(MT * 32, a.wp_rows - m0)) * WP / 8;
#pragma unroll 4
    for (int i = tid; i < nvec; i += 256) ((hx8*)Ws)[i] = i < lim ? src[i] : zero;
    for (int k = tid; k < KT * 32; k += 256) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      if (k < a.Co) { c[0] = a.coef[k * 4]; c[1] = a.coef[k * 4 + 1]; c[2] = a.coef[k * 4 + 2]; }
      *(f32x4*)&Cs[k * 4] = c;
    }
  }

  // ---- register-staged prefetch of the next tile: row (tid>>4) + 16*i, 8 points at unit (tid&15)
  const int srow = tid >> 4, sunit = tid & 15;
  hx8 rg[NVY], ry[NVY], rx[SWB ? 1 : NVX];
  // TAIL: [y > 0] of this thread's staged x vectors (8 bits each).  The epilogue owns the same (row, unit) positions as
  // the staging -- row (tid >> 4) + 16 * i, unit tid & 15 -- and rx is re-loaded with the next tile before the epilogue runs
  unsigned ymask[TAIL ? NVX : 1];
  // Every load of the tile loop is UNCONDITIONAL (clamped address; the value is zeroed / never used where the row or point
  // is outside): vmcnt retires in order and the compiler cannot count conditional accesses, so one conditional load made
  // every later wait of the iteration a vmcnt(0) -- the epilogue's wait for its own operands then also drained the next
  // tile's prefetch, and the commit's wait for the prefetch drained the epilogue's stores (tools/scan_waitcnt.py).
  auto issue = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p = (long long)(tile - n * tiles_per_n) * BN + sunit * 8;
#pragma unroll
    for (int i = 0; i < NVY; i++) {
      const int k = srow + 16 * i;
      const long long o = (k < a.Co && p < a.P) ? ((long long)n * a.Co + k) * a.P + p : 0;   // (rows >= Co: coefficients are zero)
      rg[i] = *(const hx8*)((const T*)a.g + o);
      ry[i] = *(const hx8*)((const T*)a.yraw + o);
    }
    if constexpr (!SWB) {
#pragma unroll
      for (int i = 0; i < NVX; i++) {
        const int m = m0 + srow + 16 * i;
        const long long o = (m < a.Ci && p < a.P) ? ((long long)n * a.Ci + m) * a.P + p : 0;
        rx[i] = *(const hx8*)((const T*)a.x + o);
      }
    }
  };
  auto commit = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p = (long long)(tile - n * tiles_per_n) * BN + sunit * 8;
    const bool pin = p < a.P;
#pragma unroll
    for (int i = 0; i < NVY; i++) {
      const int k = srow + 16 * i;
      if (k >= Kp) continue;                      // rows Kp.. stay zero
      const f32x4 cf = *(const f32x4*)&Cs[k * 4];   // zeros for padded rows
      const float A = pin ? cf[0] : 0.f, B = pin ? cf[1] : 0.f, C = pin ? cf[2] : 0.f;   // padded points exactly zero (C must not leak in)
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = A * (float)rg[i][e] + B * (float)ry[i][e] + C;
      VecIO<H, 8>::store(&Ys[k * YP + ((sunit ^ ((k >> 2) & 3)) << 3)], v);
    }
    if constexpr (!SWB) {
#pragma unroll
      for (int i = 0; i < NVX; i++) {
        const int m = srow + 16 * i;
        if (!(pin && m0 + m < a.Ci)) {            // zeros where m >= Ci or p >= P
#pragma unroll
          for (int e = 0; e < 8; e++) rx[i][e] = (H)0.f;
        }
        *(hx8*)&Xs[m * XP + sunit * 8] = rx[i];
        if constexpr (TAIL) {
          unsigned mk = 0;
#pragma unroll
          for (int e = 0; e < 8; e++) mk |= ((float)rx[i][e] > 0.f ? 1u : 0u) << e;
          ymask[i] = mk;
        }
      }
    }
  };

  // per-(sample, channel) sums of the SWISH_BWD epilogue
  float st1[SWB ? ROWS_PT : 1], st2[SWB ? ROWS_PT : 1];
  if constexpr (SWB) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) { st1[i] = 0.f; st2[i] = 0.f; }
  }
  // TAIL: per-channel sums of the masked gradient (kept over all tiles of the workgroup: they are not per sample)
  float tg[TAIL ? ROWS_PT : 1], tgc[TAIL ? ROWS_PT : 1], tgr[TAILR ? ROWS_PT : 1];
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) { tg[i] = 0.f; tgc[i] = 0.f; }
  }
  if constexpr (TAILR) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) tgr[i] = 0.f;
  }
  auto flush_sums = [&](int n) __attribute__((always_inline)) {
    if constexpr (SWB) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) {
        const float s1 = row16_sum(st1[i]), s2 = row16_sum(st2[i]);
        const int m = m0 + (tid >> 4) + 16 * i;
        if ((tid & 15) == 0 && m < a.Ci) {
          double* d = a.nc_sums + ((long long)n * a.Ci + m) * 2;
          atomic_add_d(d, (double)s1);
          atomic_add_d(d + 1, (double)s2);
        }
        st1[i] = 0.f;
        st2[i] = 0.f;
      }
    }
  };

  // transposed-read lane geometry (dY tile as B operand): lane -> (row 8*(g16>>1)+q (+4), 4 points)
  const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_row = 8 * (g16 >> 1) + q;
  const int tr_unit = wid * 4 + 2 * (g16 & 1) + (pp >> 1);          // logical 16-byte unit of this lane's 8 bytes
  const int tr_off0 = ((tr_unit ^ ((tr_row >> 2) & 3)) << 3) + (pp & 1) * 4;          // rows kk + tr_row
  const int tr_off1 = ((tr_unit ^ (((tr_row + 4) >> 2) & 3)) << 3) + (pp & 1) * 4;    // rows kk + tr_row + 4
  typedef s16x4_f __attribute__((address_space(3))) * lds_s16x4_ptr;

  f32x16 acc_dw[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc_dw[s][j] = 0.f;

  // dW[co][ci] += dY[co][:] . Xh[ci][:] over the 128 points of the tile
  auto wgrad_mfma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + 4 * s, kpart = 0;
      if constexpr (NKS > 1) { id = wid % NT; kpart = wid / NT; }
      if (id < NT) {
        const int cot = id / MT, cit = id - cot * MT;
        const int rowy = cot * 32 + r;
        const int swz = (rowy >> 2) & 3;
        const H* yrow = Ys + rowy * YP;
        const H* xrow = Xs + (cit * 32 + r) * XP + 8 * half;
        constexpr int KSTEPS = (BN / 16) / NKS;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
          const int u = (kpart * KSTEPS + ks) * 2 + half;
          const hx8 af = *(const hx8*)(yrow + ((u ^ swz) << 3));
          const hx8 bf = *(const hx8*)(xrow + (kpart * KSTEPS + ks) * 16);
          acc_dw[s] = mfma16<H>(af, bf, acc_dw[s]);
        }
      }
    }
  };

  // BN_b scale / shift of the rows this thread owns in the epilogue (the same for every tile: loaded once -- except in the
  // tallest panels, where sixteen more live registers spill: 48<->108 @28x28 162 -> 301 us; those reload them per tile)
  constexpr bool COEF_ONCE = MT <= 2;
  float esb[SWB ? ROWS_PT : 1], etb[SWB ? ROWS_PT : 1];
  if constexpr (SWB && COEF_ONCE) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) {
      const int m = m0 + (tid >> 4) + 16 * i;
      esb[i] = m < a.Ci ? a.b_ss[m * 2] : 0.f;
      etb[i] = m < a.Ci ? a.b_ss[m * 2 + 1] : 0.f;
    }
  }

  if (tile_begin < tile_end) issue(tile_begin);
  int n_prev = tile_begin < tile_end ? tile_begin / tiles_per_n : 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    __syncthreads();            // every reader of the previous tile's dY / Xh / slab is done
    commit(tile);
    __syncthreads();
    // (the per-sample flush -- conditional atomics the compiler cannot count -- sits BEHIND the commit: in front of it the
    // commit's wait for the prefetched tile was a vmcnt(0) that also drained the previous tile's stores)
    if constexpr (SWB) {
      if (n != n_prev) flush_sums(n_prev);
    }
    n_prev = n;

    // ---- global loads of this tile's epilogue, THEN the next tile's prefetch: the epilogue's wait (vmcnt(#prefetch loads))
    // leaves the prefetch in flight, and the matrix-core phase below covers the epilogue operands' latency
    const int oc = (tid & 15) * 8;
    constexpr bool EPL8 = (EPI == X3D_EPI_ADD) || SWB;
    hx8 epl8[EPL8 ? ROWS_PT : 1];
    hx4 epl4[EPL8 ? 1 : ROWS_PT];
    hx8 tc8[TAIL ? ROWS_PT : 1], tr8[TAILR ? ROWS_PT : 1];
    float egt[SWB ? ROWS_PT : 1];
    constexpr bool epl4_vec = E4V;   // rows of 8k points: the strided shortcut gradient as 8-byte loads
    // (the tallest panels keep the loads next to their use: hoisting eight rows' worth of registers spills -- and so does the
    // 2 x 4 panel of the `c` convs, whose hoisted braw rows live across both matrix-core phases: 48<->108 @28x28 162 -> 301 us)
    constexpr bool HOIST = MT <= 2 && !(SWB && MT * KT >= 8);
    auto epi_load = [&](int i) __attribute__((always_inline)) {
      const int m = m0 + (tid >> 4) + 16 * i;
      const long long p = p0 + oc;
      const bool ok = m < a.Ci && p < a.P;
      // HOIST: unconditional loads from a clamped address (countable: see issue()).  The tall panels load next to the use,
      // inside the slab loop, and keep the load CONDITIONAL: without the branch the scheduler hoists all eight rows' loads to
      // the top of the unrolled loop and the panel spills 300 bytes per lane (48<->108 @28x28: 162 -> 301 us)
      const bool ld = HOIST || ok;
      const long long orow = ok ? ((long long)n * a.Ci + m) * a.P + p : 0;
      if constexpr (TAIL) {
#pragma unroll
        for (int e = 0; e < 8; e++) tc8[i][e] = (H)0.f;
        if constexpr (TAILR) {
#pragma unroll
          for (int e = 0; e < 8; e++) tr8[i][e] = (H)0.f;
        }
        if (ld) {
          const hx8 lc = *(const hx8*)((const T*)a.tail_c + orow);
          if (ok) tc8[i] = lc;
          if constexpr (TAILR) {
            const hx8 lr = *(const hx8*)((const T*)a.tail_r + orow);
            if (ok) tr8[i] = lr;
          }
        }
      }
      if constexpr (EPL8) {
#pragma unroll
        for (int e = 0; e < 8; e++) epl8[i][e] = (H)0.f;
        if (ld) {
          const hx8 l8 = *(const hx8*)((const T*)(SWB ? a.braw : a.add) + orow);
          if (ok) epl8[i] = l8;
        }
        if constexpr (SWB) {
          if constexpr (!COEF_ONCE) {
            esb[i] = m < a.Ci ? a.b_ss[m * 2] : 0.f;
            etb[i] = m < a.Ci ? a.b_ss[m * 2 + 1] : 0.f;
          }
          const bool okg = m < a.Ci && a.egate;
          egt[i] = 1.0f;
          if (HOIST || okg) {
            const float gl = (a.egate ? a.egate : a.b_ss)[okg ? (long long)n * a.Ci + m : 0];
            if (okg) egt[i] = gl;
          }
        }
      } else {
        const int hw = a.eH * a.eW;      // per-sample point counts fit 32 bits (host check): 32-bit divisions
        const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
        const int T_ = (int)a.P / hw;
        const int t = (int)p / hw;
        const int rem = (int)p - t * hw;
        const int h = rem / a.eW, w = rem - h * a.eW;
        const bool okv = ok && epl4_vec && (h & 1) == 0;
#pragma unroll
        for (int e = 0; e < 4; e++) epl4[i][e] = (H)0.f;
        if (HOIST || okv) {
          const hx4 l4 = *(const hx4*)((const T*)a.add + (okv ? ((((long long)n * a.Ci + m) * T_ + t) * Hh + (h >> 1)) * Wh + (w >> 1) : 0));
          if (okv) epl4[i] = l4;
        }
      }
    };
    if constexpr (HOIST) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) epi_load(i);
    }
    issue(min(tile + 1, tile_end - 1));   // (past the end: the last tile again -- the number of loads in flight stays static)

    // ---- dX tile: acc[s] (rows s*32.., this wave's 32 points) = W^T dY
    f32x16 acc[MT];
#pragma unroll
    for (int s = 0; s < MT; s++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[s][j] = 0.f;
    for (int kk = 0; kk < Kp; kk += 16) {
      const s16x4_f b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Ys[(kk + tr_row) * YP + tr_off0]));
      const s16x4_f b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Ys[(kk + tr_row + 4) * YP + tr_off1]));
      const s16x8_f bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      const hx8 bfrag = __builtin_bit_cast(hx8, bs);
#pragma unroll
      for (int s = 0; s < MT; s++) {
        const hx8 afrag = *(const hx8*)&Ws[(s * 32 + r) * WP + kk + 8 * half];
        acc[s] = mfma16<H>(afrag, bfrag, acc[s]);
      }
    }
    if constexpr (!SWB) wgrad_mfma();   // Xh came with the tile: finish with the dY tile before the slab reuses its LDS

    // ---- epilogue in 32-row slabs through LDS: thread owns rows (tid>>4) + 16*ii, 8 points at (tid&15)*8
    // dx stores are bounds-checked buffer stores (an offset past the sample's [Ci][P] matrix is dropped): a static number of
    // stores per tile, which the commit's wait for the prefetched tile can count instead of draining
    __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc((T*)a.dx + (long long)n * a.Ci * a.P, 0,
                                                                   (int)((long long)a.Ci * a.P * 2), 0x00020000);
#pragma unroll
    for (int sl = 0; sl < MT; sl++) {
      if constexpr (!HALF_SLAB) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) Os[((j & 3) + 8 * (j >> 2) + 4 * half) * OP + wid * 32 + r] = acc[sl][j];
        __syncthreads();
      }
#pragma unroll
      for (int ii = 0; ii < 2; ii++) {
        if constexpr (HALF_SLAB) {   // rows 16*ii .. 16*ii+15 of the tile are accumulator registers 8*ii .. 8*ii+7
          __syncthreads();
#pragma unroll
          for (int j = 0; j < 8; j++) Os[((j & 3) + 8 * (j >> 2) + 4 * half) * OP + wid * 32 + r] = acc[sl][8 * ii + j];
          __syncthreads();
        }
        const int i = sl * 2 + ii;
        const int row = HALF_SLAB ? (tid >> 4) : (tid >> 4) + 16 * ii;   // row inside the slab
        const int ml = sl * 32 + (tid >> 4) + 16 * ii;   // row inside this workgroup's slice
        const int m = m0 + ml;
        const long long p = p0 + oc;
        // no branch around the row: rows / points outside run the arithmetic on zeros (their accumulators, staged operands
        // and coefficients are zero) and their store is dropped by the buffer bounds check
        const bool rvalid = m < a.Ci && p < a.P;
        if constexpr (!HOIST) { if (!rvalid) continue; }   // (the tall panels keep the branchy form: see epi_load)
        float val[8];
        {
          const f32x4 v0 = *(const f32x4*)&Os[row * OP + oc], v1 = *(const f32x4*)&Os[row * OP + oc + 4];
#pragma unroll
          for (int e = 0; e < 4; e++) { val[e] = v0[e]; val[4 + e] = v1[e]; }
        }
        if constexpr (!HOIST) epi_load(i);
        if constexpr (EPI == X3D_EPI_ADD) {
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] += (float)epl8[i][e];
        } else if constexpr (EPI == X3D_EPI_ADD_STRIDED) {
          const int hw = a.eH * a.eW;      // per-sample point counts fit 32 bits (host check): 32-bit divisions
          const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
          const int T_ = (int)a.P / hw;
          if (epl4_vec) {   // loaded above (zeros on odd rows)
#pragma unroll
            for (int e = 0; e < 4; e++) val[2 * e] += (float)epl4[i][e];
          } else if (rvalid) {
            for (int e = 0; e < 8; e++) {
              const int pe = (int)p + e;
              const int t = pe / hw;
              const int rem = pe - t * hw;
              const int h = rem / a.eW, w = rem - h * a.eW;
              if (((h | w) & 1) == 0) {
                const long long oa = ((((long long)n * a.Ci + m) * T_ + t) * Hh + (h >> 1)) * Wh + (w >> 1);
                val[e] += to_f<T>(((const T*)a.add)[oa]);
              }
            }
          }
        } else if constexpr (SWB) {
          float b[8], xh[8];
#pragma unroll
          for (int e = 0; e < 8; e++) b[e] = (float)epl8[i][e];
          const SwishCoef sc_ = swish_coef(esb[i], etb[i], egt[i]);
#pragma unroll
          for (int e = 0; e < 8; e++) {
            float d_;
            swish_bwd_(sc_, b[e], xh[e], d_);                      // xh: conv input of the forward pass
            const float dv = val[e] * d_;
            val[e] = dv;
            st1[i] += dv;
            st2[i] += dv * b[e];
          }
          VecIO<H, 8>::store(&Xs[ml * XP + oc], xh);
        }
        if constexpr (TAIL) {   // Add + ReLU backward of the block this gradient leaves: mask, then the BN_c / BN_r backward sums
          const unsigned mk = ymask[i];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float gm = ((mk >> e) & 1u) ? round_to<T>(val[e]) : 0.f;   // the sums describe dx as stored (as x3d_tail_bwd's do)
            val[e] = gm;
            tg[i] += gm;
            tgc[i] += gm * (float)tc8[i][e];
            if constexpr (TAILR) tgr[i] += gm * (float)tr8[i][e];
          }
        }
        {
          hx8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) ov[e] = (H)val[e];
          typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, ov), dxr,
                                                 rvalid ? (unsigned)(((long long)m * a.P + p) * 2) : 0x80000000u, 0, 0);
        }
      }
    }
    if constexpr (SWB) {
      __syncthreads();   // Xh tile complete
      wgrad_mfma();
    }
  }
  if (tile_begin < tile_end) flush_sums(n_prev);

  if constexpr (TAIL) {
    if (tile_begin < tile_end) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) {
        const float s0 = row16_sum(tg[i]), s1 = row16_sum(tgc[i]);
        float s2 = 0.f;
        if constexpr (TAILR) s2 = row16_sum(tgr[i]);
        const int m = m0 + (tid >> 4) + 16 * i;
        if ((tid & 15) == 0 && m < a.Ci) {
          atomic_add_d(&a.tail_sums_c[m * 2], (double)s0);
          atomic_add_d(&a.tail_sums_c[m * 2 + 1], (double)s1);
          if constexpr (TAILR) {
            atomic_add_d(&a.tail_sums_r[m * 2], (double)s0);
            atomic_add_d(&a.tail_sums_r[m * 2 + 1], (double)s2);
          }
        }
      }
    }
  }

  // ---- dW partial -> global (fp32 atomics)
  if (tile_begin < tile_end) {
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + 4 * s;
      if constexpr (NKS > 1) id = wid % NT;
      if (id < NT) {
        const int cot = id / MT, cit = id - cot * MT;
        const int ci = m0 + cit * 32 + r;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int co = cot * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
          if (co < a.Co && ci < a.Ci) atomicAdd(&a.dw[(long long)co * a.Ci + ci], acc_dw[s][j]);
        }
      }
    }
  }
}

static inline size_t fb_lds_bytes(int MT, int KT, int Kp, bool swb) {
  return (size_t)KT * 32 * FB_YP * 2 + (size_t)MT * 32 * FB_XP * 2 + (size_t)MT * 32 * (Kp + 8) * 2 +
         (swb ? (size_t)(MT == 4 ? 16 : 32) * FB_OP * 4 : 0) + (size_t)KT * 32 * 16;
}

template <typename H, int MT, int KT, int EPI, int TAIL = 0, bool E4V = true>
static int fb_launch(PwBwdArgs& a, hipStream_t st) {
  if constexpr (EPI == X3D_EPI_ADD_STRIDED && E4V) {
    if ((a.eW & 7) != 0) return fb_launch<H, MT, KT, EPI, TAIL, false>(a, st);
  }
  const size_t lds = fb_lds_bytes(MT, KT, a.Kp, EPI == X3D_EPI_SWISH_BWD);
  X3D_REQUIRE(lds <= 160 * 1024, "pw_bwd: needs %zu B of LDS", lds);
  if constexpr (!E4V) X3D_DESCRIBE("pw_bwd_fused_kernel<%s, %d, %d, %d, %d, e>", HV<H>::name, MT, KT, EPI, TAIL);
  else if constexpr (TAIL != 0) X3D_DESCRIBE("pw_bwd_fused_kernel<%s, %d, %d, %d, %d>", HV<H>::name, MT, KT, EPI, TAIL);
  else X3D_DESCRIBE("pw_bwd_fused_kernel<%s, %d, %d, %d>", HV<H>::name, MT, KT, EPI);
  auto kern = pw_bwd_fused_kernel<H, MT, KT, EPI, TAIL, E4V>;
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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    slots = nb * cus;
    if (occ_n < 8) { occ_lds[occ_n] = lds; occ_slots[occ_n] = slots; occ_n++; }
  }
  const long long total_tiles = ceil_div_ll(a.P, FB_BN) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_bwd: too many tiles");
  const int gy = ceil_div(ceil_div(a.Ci, 32), MT);    // slices of MT*32 input channels
  long long tpb = ceil_div_ll(total_tiles * gy, slots);   // one balanced round
  if (tpb < 4) tpb = 4;                               // keeps the dW atomics (<= 32 KB) small against the streamed tiles
  a.tiles_per_block = (int)tpb;
  long long gx = ceil_div_ll(total_tiles, tpb);
  // workgroups go to the 8 XCDs round-robin by linear index x + gx * y: with gx a multiple of 8 the slices (blockIdx.y)
  // of one point-tile range share an XCD -- and its L2 -- so the dY / yraw tiles they all stage are fetched once
  // (surplus workgroups find no tiles and leave)
  if (gy > 1 && xcd_pad_enabled()) gx = (gx + 7) & ~7ll;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_bwd_fused");
  return X3D_OK;
}

// panel shape: KT covers all of Co (<= 128); MT covers Ci when MT*KT <= 8 dW tiles fit the accumulators, else Ci is
// sliced over blockIdx.y in panels of MT = 8 / KT row tiles
static inline bool fb_shape(int Ci, int Co, int* MT, int* KT) {
  const int mt = ceil_div(Ci, 32), kt = ceil_div(Co, 32);
  if (kt > 4) return false;
  *KT = kt;                                     // 1..4 (3: 96-channel dY tiles keep two workgroups per CU)
  const int cap = 8 / *KT;                      // 8, 4, 2, 2 row tiles
  int m = mt <= 1 ? 1 : (mt == 2 ? 2 : 4);
  if (m > cap) m = cap;
  if (m > 4) m = 4;
  *MT = m;
  return true;
}

template <typename H, int EPI>
static int fb_pick(PwBwdArgs& a, hipStream_t st) {
  int MT = 0, KT = 0;
  if (!fb_shape(a.Ci, a.Co, &MT, &KT)) { x3d_set_error("pw_bwd: unsupported tile shape"); return X3D_ERR_INVALID; }
  if constexpr (EPI != X3D_EPI_SWISH_BWD) {
    if (a.tail_c) {          // fb_supported: only panels of one or two row tiles carry the tail
#define FB_TCASE(M_, K_) if (MT == M_ && KT == K_) return a.tail_r ? fb_launch<H, M_, K_, EPI, 2>(a, st) : fb_launch<H, M_, K_, EPI, 1>(a, st);
      FB_TCASE(1, 1) FB_TCASE(1, 2) FB_TCASE(1, 3) FB_TCASE(1, 4) FB_TCASE(2, 2)
#undef FB_TCASE
      x3d_set_error("pw_bwd: no tail instantiation for this panel");
      return X3D_ERR_INVALID;
    }
  }
#define FB_CASE(M_, K_) if (MT == M_ && KT == K_) return fb_launch<H, M_, K_, EPI>(a, st);
  FB_CASE(1, 1) FB_CASE(1, 2) FB_CASE(1, 3) FB_CASE(1, 4) FB_CASE(2, 1) FB_CASE(2, 2) FB_CASE(2, 3) FB_CASE(2, 4)
  FB_CASE(4, 1) FB_CASE(4, 2)
#undef FB_CASE
  x3d_set_error("pw_bwd: unsupported tile shape");
  return X3D_ERR_INVALID;
}

// the weights-stationary fused backward of the stage-4 `c` conv (pw_bwd_wst.hip)
bool pw_bwd_wst_applies(const x3d_pw_bwd_args* b);
int pw_bwd_wst(const x3d_pw_bwd_args* b, hipStream_t st);
int pw_bwd_wst_dw_parts(const x3d_pw_bwd_args* b);
// ... and of the stage-4 `a` conv (pw_bwd_wsta.hip)
bool pw_bwd_wsta_applies(const x3d_pw_bwd_args* b);
int pw_bwd_wsta(const x3d_pw_bwd_args* b, hipStream_t st);
int pw_bwd_wsta_dw_parts(const x3d_pw_bwd_args* b);

// ... and the form that recomputes the conv output instead of reading it (pw_bwd_rc.hip; rc_panel != NULL)
bool pw_bwd_rc_supported(const x3d_pw_bwd_args* b);
int pw_bwd_rc(const x3d_pw_bwd_args* b, hipStream_t st);

// eligibility of the fused path (the caller falls back to x3d_pw_dgrad + x3d_pw_wgrad otherwise)
static bool fb_supported(const x3d_pw_bwd_args* b) {
  if (b->rc_panel) return pw_bwd_rc_supported(b);
  if (pw_bwd_wst_applies(b) || pw_bwd_wsta_applies(b)) return true;
  if (!x3d_is_half(b->dtype) || !b->w_panel || !b->coef || !b->yraw) return false;
  int MT = 0, KT = 0;
  if (!fb_shape(b->Cin, b->Cout, &MT, &KT)) return false;
  // slicing Ci re-stages the dY tile once per slice: worth it up to ~4 slices (x3d_pw_dgrad does the same per row block)
  if (ceil_div(ceil_div(b->Cin, 32), MT) > 4) return false;
  const long long P = (long long)b->T * b->H * b->W;
  if (P % 8 || P >= (1ll << 31)) return false;      // 32-bit point indices in the kernel
  if ((long long)b->Cin * P * 2 >= (1ll << 31)) return false;   // one sample's dx inside the 2 GB buffer-store window
  const void* ps[] = {b->g, b->yraw, b->dx, b->w_panel, b->epi == X3D_EPI_SWISH_BWD ? b->braw : b->x,
                      b->epi == X3D_EPI_ADD ? b->add : nullptr};
  for (const void* p : ps) if (p && ((uintptr_t)p % 16)) return false;
  if (b->epi == X3D_EPI_ADD_STRIDED && ((uintptr_t)b->add % 8)) return false;
  if (b->epi != X3D_EPI_ADD && b->epi != X3D_EPI_ADD_STRIDED && b->epi != X3D_EPI_SWISH_BWD) return false;
  if (b->tail_c) {     // the folded residual-tail backward: `a`-conv epilogues, panels of <= 2 row tiles, one slice
    // panels whose tail instantiation stays inside the register file (r03e: the 2x3 / 2x4 panels spill 52-240 bytes per
    // lane with the two extra epilogue operands and lose more than the separate tail pass costs: 108<->48 @28x28 128 -> 209 us)
    if (b->epi == X3D_EPI_SWISH_BWD || ceil_div(ceil_div(b->Cin, 32), MT) > 1) return false;
    if (!(MT == 1 || (MT == 2 && KT == 2))) return false;
    if (((uintptr_t)b->tail_c % 16) || (b->tail_r && ((uintptr_t)b->tail_r % 16))) return false;
  }
  const int Kp = (b->Cout + 15) & ~15;
  return fb_lds_bytes(MT, KT, Kp, b->epi == X3D_EPI_SWISH_BWD) <= 160 * 1024;
}

extern "C" int x3d_pw_bwd_supported(const x3d_pw_bwd_args* b) { return (b && fb_supported(b)) ? 1 : 0; }

// partial-slab form of the weight-gradient flush (x3d_hip.h dw_slab): the persistent weights-stationary kernels only
extern "C" int x3d_pw_bwd_dw_parts(const x3d_pw_bwd_args* b) {
  if (!b || b->rc_panel || b->N <= 0 || b->T <= 0 || b->H <= 0 || b->W <= 0) return 0;
  if (pw_bwd_wst_applies(b)) return pw_bwd_wst_dw_parts(b);
  if (pw_bwd_wsta_applies(b)) return pw_bwd_wsta_dw_parts(b);
  return 0;
}

extern "C" int x3d_pw_bwd(const x3d_pw_bwd_args* b, void* stream) {
  X3D_REQUIRE(b && b->g && b->dx, "pw_bwd: null pointer");
  X3D_REQUIRE(b->N > 0 && b->Cin > 0 && b->Cout > 0 && b->T > 0 && b->H > 0 && b->W > 0, "pw_bwd: bad extents");
  X3D_REQUIRE(!b->dw_slab || (x3d_pw_bwd_dw_parts(b) > 0 && ((uintptr_t)b->dw_slab % 16) == 0),
              "pw_bwd: dw_slab given but the kernel behind this call has no slab form (x3d_pw_bwd_dw_parts() == 0)");
  // the grid of the slab kernels is x3d_pw_bwd_dw_parts() (the same function of the device's CU count): a buffer sized for
  // another count would be overrun, or summed with unwritten slabs
  X3D_REQUIRE(!b->dw_slab || x3d_describe.out || x3d_pw_bwd_dw_parts(b) == b->dw_slab_parts,
              "pw_bwd: dw_slab holds %d slabs, this launch writes %d (x3d_pw_bwd_dw_parts)", b->dw_slab_parts, x3d_pw_bwd_dw_parts(b));
  X3D_REQUIRE(x3d_describe.out || bn_bwd_fold_ok(b->coef_fold), "pw_bwd: incomplete coef_fold");
  X3D_REQUIRE(!b->coef_fold || (!b->rc_panel && (pw_bwd_wst_applies(b) || pw_bwd_wsta_applies(b))),
              "pw_bwd: coef_fold is not taken by the kernel behind this call (x3d_pw_coef_fold_supported() == 0)");
  if (b->rc_panel) return pw_bwd_rc(b, (hipStream_t)stream);
  X3D_REQUIRE(b->yraw && (b->coef || b->coef_fold) && b->dw, "pw_bwd: null pointer");
  X3D_REQUIRE(fb_supported(b), "pw_bwd: shape / alignment / epilogue not covered by the fused kernel "
                               "(x3d_pw_bwd_supported() == 0): use x3d_pw_dgrad + x3d_pw_wgrad");
  if (pw_bwd_wst_applies(b)) return pw_bwd_wst(b, (hipStream_t)stream);
  if (pw_bwd_wsta_applies(b)) return pw_bwd_wsta(b, (hipStream_t)stream);
  if (b->epi == X3D_EPI_SWISH_BWD)
    X3D_REQUIRE(b->braw && b->b_scale_shift && b->nc_sums, "pw_bwd: SWISH_BWD needs braw/b_scale_shift/nc_sums");
  else
    X3D_REQUIRE(b->x && b->add, "pw_bwd: ADD epilogues need x (conv input) and add");
  X3D_REQUIRE(!b->tail_c || (b->tail_sums_c && (!b->tail_r || b->tail_sums_r)) || x3d_describe.out, "pw_bwd: tail_c / tail_r need their sums");
  X3D_REQUIRE(b->tail_c || !b->tail_r, "pw_bwd: tail_r without tail_c");
  PwBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g = b->g; a.yraw = b->yraw; a.coef = b->coef;
  a.wp = b->w_panel; a.wp_rows = (b->Cin + 31) & ~31;
  a.dx = b->dx; a.add = b->add; a.braw = b->braw; a.b_ss = b->b_scale_shift; a.egate = b->gate; a.nc_sums = b->nc_sums;
  a.eH = b->H; a.eW = b->W;
  a.x = b->x; a.dw = b->dw;
  a.tail_c = b->tail_c; a.tail_r = b->tail_r; a.tail_sums_c = b->tail_sums_c; a.tail_sums_r = b->tail_sums_r;
  a.N = b->N; a.Co = b->Cout; a.Ci = b->Cin; a.Kp = (b->Cout + 15) & ~15;
  a.P = (long long)b->T * b->H * b->W;
  hipStream_t st = (hipStream_t)stream;
  if (b->dtype == X3D_F16) {
    switch (b->epi) {
      case X3D_EPI_ADD: return fb_pick<f16, X3D_EPI_ADD>(a, st);
      case X3D_EPI_ADD_STRIDED: return fb_pick<f16, X3D_EPI_ADD_STRIDED>(a, st);
      default: return fb_pick<f16, X3D_EPI_SWISH_BWD>(a, st);
    }
  }
  switch (b->epi) {
    case X3D_EPI_ADD: return fb_pick<bf16, X3D_EPI_ADD>(a, st);
    case X3D_EPI_ADD_STRIDED: return fb_pick<bf16, X3D_EPI_ADD_STRIDED>(a, st);
    default: return fb_pick<bf16, X3D_EPI_SWISH_BWD>(a, st);
  }
}
