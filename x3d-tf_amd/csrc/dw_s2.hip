// Fused backward of the channelwise 3x3x3 convolution, STRIDE 2, strips of two outputs, rows that are whole aligned staging
// vectors in 16-bit storage (X3D-M: 112x112 -> 56x56 and 56x56 -> 28x28, the launches with the largest total of the train step).
//
// Why a kernel of its own.  dw3d_bwd_pd_kernel<T, 2, 2, 8, 2> (dw_pd.hip) is bound by VALU ISSUE, not by HBM: with every global
// access out of range it keeps 77 % of its time, with the tap sums skipped as well still 53 % (X3D_DW_PD_EXP, experiments
// build; profiles/r05_ab_dw_s2.txt), and SQ_INSTS_VALU fills 79 % of the SIMDs' issue slots.  Per thread and plane (two
// outputs, a 2 x 4 block of the data gradient) it issues ~286 vector instructions of which 108 are the multiply-adds; the
// rest is staging, the emit, address arithmetic and register copies.  Here: ~165.
//   * register ROLES instead of copies (dw3d_bwd_pd_s1_kernel's scheme): the T loop is unrolled by 6, so "the previous plane"
//     (period 2) and "the accumulator of plane t-1 / t / t+1" (period 3) are compile-time facts: -45 v_mov per plane; an
//     accumulator is never zeroed either -- the first taps a plane receives are written as products;
//   * plane offsets in the SCALAR offset of the buffer instructions (it takes part in the range check like the vector one): the
//     per-thread offsets are loop constants, -9 v_add per plane and ~25 VGPRs (162 -> 134 in the two-barrier form);
//   * the data gradient on PACKED FMAs: the 2 x 4 block is four column pairs; (dA[2i], dA[2i+1]) += (w[kh][0], w[kh][1]) * dB
//     with the weight pair in an SGPR pair and dB broadcast by op_sel, the lone third tap as a scalar FMA into the even
//     column -- 3 packed + 3 scalar instructions where the copying kernel had 9 FMAs with an SGPR operand and 4 adds
//     (scalar FMAs instead: + 3 % time; packed staging / packed sums alone: no difference);
//   * the ReLU mask of the emit from the WINDOW: the block a thread owns (rows 2 ho, 2 ho + 1, columns 2 wo0 .. 2 wo0 + 3) is
//     the upper-left 2 x 4 corner of the activation window it read for the same plane (TF-SAME with pw = 0):
//     relu(sc a + sh) > 0 needs no FMA, and rows outside the image are zero rows of the LDS plane; per-channel sums on packed
//     add / FMA, one v_cvt_pk per stored pair; the araw block for the sum of ga * araw is loaded one iteration before its emit
//     (its lines are in L2 since the staging load) instead of travelling through two register copies.
// Two forms.  dw3d_bwd_s2_kernel: the LDS planes and two barriers per plane of the kernel it replaces (134 VGPRs, three waves per
// SIMD); any plane size.  dw3d_bwd_s2r_kernel (default when the planes fit its constant buffer strides): an LDS RING, below.
// Measured, 64 clips, isolated launches (tools/ab_dw.py, alternating): 54 ch 112 -> 56: 970 -> 880 (two barriers) -> 856 us (ring);
// 108 ch 56 -> 28: 555 -> 497 -> 485; in the train step 20.51 -> 20.34 ms (the two launches average 622 -> 523 us there).
// What did NOT pay (same harness): one barrier per plane with two LDS buffers at three waves (+ 3 %), three planes in flight
// instead of two (+- 0), four waves per SIMD with 24 B of spills in the loop (1175 us: a spill is a vector-memory operation in
// the in-order vmcnt queue).  Sums are taken in a different order than in dw3d_bwd_pd_kernel (tests: fp64 reference).
#include "dw_common.h"

#ifndef S2_OCC
#define S2_OCC 3
#endif
template <typename T, int CV, int PD, int UN>
__global__ __launch_bounds__(256, S2_OCC) void dw3d_bwd_s2_kernel(const DwBwdArgs a) {
  constexpr int S = 2, SW = 2, WIN = 5, BW = 3, NA = 4, NR = 2;
  constexpr int VA = CV, VB = CV / 2, EB = (int)sizeof(T);
  static_assert(UN % 6 == 0 && UN % PD == 0, "roles have periods 2 (windows) and 3 (planes); slots period PD");
  static_assert(VA % 2 == 0 && VB % 2 == 0 && sizeof(T) == 2, "staging in pairs, 16-bit storage");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  const int aplane = g.RIN * g.LP;
  const int bplane = a.RB * a.LPB;
  float* Al = lds;
  float* Bl = lds + aplane;
  float* scratch = Bl + bplane;

  int b = blockIdx.x;
  const int tile = __builtin_amdgcn_readfirstlane(b % g.ntile_h); b /= g.ntile_h;
  const int c = __builtin_amdgcn_readfirstlane(b % g.C);
  const int n = __builtin_amdgcn_readfirstlane(b / g.C);
  const int h0 = tile * g.TH;
  const int th_here = min(g.TH, g.Ho - h0);
  const int r = threadIdx.x / g.nstrips, sidx = threadIdx.x - r * g.nstrips;
  const bool active = r < th_here;
  const int ho = h0 + r, wo0 = sidx * SW;

  for (int i = threadIdx.x; i < aplane + bplane; i += blockDim.x) lds[i] = 0.f;

  float wgt[27];
#pragma unroll
  for (int k = 0; k < 27; k++) wgt[k] = a.w[c * 27 + k];
  const float sc = a.ss_a[c * 2], sh = a.ss_a[c * 2 + 1];
  const float* cf = a.coef_nc + ((long long)n * g.C + c) * 4;
  const float cA = cf[0], cB = cf[1], cC = cf[2];

  const int iplB = g.H * g.W * EB, oplB = g.Ho * g.Wo * EB;
  const long long chan = (long long)n * g.C + c;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.araw + chan * g.T * g.H * g.W, 0, g.T * iplB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.ga + chan * g.T * g.H * g.W, 0, g.T * iplB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.dv + chan * g.T * g.Ho * g.Wo, 0, g.T * oplB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.braw + chan * g.T * g.Ho * g.Wo, 0, g.T * oplB, 0x00020000);
  const int rowA0 = h0 * S - g.ph;

  // staging maps: ONE vector per thread and tensor (host guarantees it); pw == 0, so LDS column = image column
  int gA = DW_OOB, lA = 0, gB = DW_OOB, lB = 0;
  bool okA = false, okB = false;
  {
    const int nvr = g.W / VA, v = threadIdx.x;
    if (v < g.RIN * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = rowA0 + lr;
      if (hi >= 0 && hi < g.H) { okA = true; gA = (hi * g.W + jv * VA) * EB; lA = lr * g.LP + jv * VA; }
    }
  }
  {  // dB plane: lds row 0 <-> output row h0-1, col 0 <-> col -1
    const int nvr = g.Wo / VB, v = threadIdx.x;
    if (v < a.RB * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = h0 - 1 + lr;
      if (hi >= 0 && hi < g.Ho) { okB = true; gB = (hi * g.Wo + jv * VB) * EB; lB = lr * a.LPB + 1 + jv * VB; }
    }
  }
  // the 2 x 4 block of dA this thread owns: rows hA, hA + 1 = window rows 0, 1; columns 2 wo0 .. 2 wo0 + 3 = window columns 0 .. 3
  const int hA = ho * 2 - g.ph, wA0 = wo0 * 2;
  int oOwn[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int h = hA + q;
    oOwn[q] = (active && h >= 0 && h < g.H) ? (h * g.W + wA0) * EB : DW_OOB;
  }
#ifdef X3D_EXPERIMENTS
  const bool x_nomath = a.exp & 1, x_noload = a.exp & 2, x_noemit = a.exp & 4;
  if (x_noload) { gA = DW_OOB; gB = DW_OOB; oOwn[0] = DW_OOB; oOwn[1] = DW_OOB; }
#else
  constexpr bool x_nomath = false, x_noload = false, x_noemit = false;
#endif

  // Plane offsets travel in the SCALAR offset of the buffer instructions (it takes part in the range check: a plane past T, or
  // DW_OOB for a dropped store, moves nothing) -- the per-thread vector offsets are loop constants, no address arithmetic per plane.
  Raw sA[PD], sD[PD], sR[PD];   // slot p % PD: the staging vectors (araw, dv, b_raw) of plane p
  Raw ownr[2][NR];              // [p & 1]: the araw block of plane p (loaded in iteration p -- its lines are in L2 since the staging load -- for the emit of iteration p + 1)
  auto issue = [&](int t, int sl, int par) {
    raw_bload<VA * EB>(sA[sl], rsA, gA, (t + PD) * iplB);
    raw_bload<VB * EB>(sD[sl], rsD, gB, (t + PD) * oplB);
    raw_bload<VB * EB>(sR[sl], rsR, gB, (t + PD) * oplB);
#pragma unroll
    for (int q = 0; q < NR; q++) raw_bload<NA * EB>(ownr[par][q], rsA, oOwn[q], t * iplB);
  };

  v2f dAr[3][NR][SW];          // dAr[p % 3][row][column pair] = gradient plane p while it is being accumulated.  Never zeroed: the
#pragma unroll                 // first taps a plane receives (kt = 2, from dB plane p - 1) are written as products, not added
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int q = 0; q < NR; q++)
#pragma unroll
      for (int i = 0; i < SW; i++) dAr[k][q][i] = (v2f){0.f, 0.f};
  float dW[27];
#pragma unroll
  for (int k = 0; k < 27; k++) dW[k] = 0.f;
  float winA2[2][3][WIN], dBs[2][SW];   // [t % 2]: this plane's act window / own dB strip; [1 - t % 2]: the previous plane's
#pragma unroll
  for (int k = 0; k < 2; k++) {
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
      for (int j = 0; j < WIN; j++) winA2[k][kh][j] = 0.f;
#pragma unroll
    for (int i = 0; i < SW; i++) dBs[k][i] = 0.f;
  }
  v2f s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int q = 0; q < NR; q++) ownr[k][q].w[0] = ownr[k][q].w[1] = ownr[k][q].w[2] = ownr[k][q].w[3] = 0u;

  // plane p is complete: mask with ReLU'(BN_a(a)) read off the window of plane p (wm = relu(sc a + sh), zero where the row is
  // outside the image or the thread idle), per-channel sums, store.  !live (p < 0): the window of "plane -1" is zero -- the
  // mask is all false, nothing is summed, the store is dropped.
  auto emit = [&](int t, bool live, const v2f (&v)[NR][SW], const float (&wm)[3][WIN], const Raw (&own)[NR]) {
    const int off = (live && !x_noload) ? t * iplB : DW_OOB;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      Raw o;
#pragma unroll
      for (int j = 0; j < SW; j++) {
        v2f gp, ap = {raw_get<T>(own[q], 2 * j), raw_get<T>(own[q], 2 * j + 1)};
        if (x_noemit) gp = v[q][j];
        else {
          gp.x = wm[q][2 * j] > 0.f ? v[q][j].x : 0.f;
          gp.y = wm[q][2 * j + 1] > 0.f ? v[q][j].y : 0.f;
          s1p += gp;
          s2p = pk_fma(gp, ap, s2p);
        }
        o.w[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(gp, typename HV<T>::x2));   // one v_cvt_pk per pair
      }
      raw_bstore<NA * EB>(o, rsG, oOwn[q], off);
    }
  };
  const v2f sc2 = bc2(sc), sh2 = bc2(sh), cA2 = bc2(cA), cB2 = bc2(cB), cC2 = bc2(cC);
  auto stage = [&](int sl) {   // slot sl -> the LDS planes (BN_a + ReLU; the BatchNorm-backward combination)
    if (okA) {
      float* dst = Al + lA;
#pragma unroll
      for (int e = 0; e < VA; e += 2) {
        v2f z = pk_fma((v2f){raw_get<T>(sA[sl], e), raw_get<T>(sA[sl], e + 1)}, sc2, sh2);
        z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f);
        *(v2f*)(dst + e) = z;
      }
    }
    if (okB) {
      float* dst = Bl + lB;
#pragma unroll
      for (int e = 0; e < VB; e += 2) {
        const v2f u = pk_fma((v2f){raw_get<T>(sR[sl], e), raw_get<T>(sR[sl], e + 1)}, cB2, cC2);
        const v2f z = pk_fma((v2f){raw_get<T>(sD[sl], e), raw_get<T>(sD[sl], e + 1)}, cA2, u);
        dst[e] = z.x; dst[e + 1] = z.y;
      }
    }
  };

  // prologue: the same load / store sequence as a steady-state iteration (stores dropped)
#pragma unroll
  for (int d = 0; d < PD; d++) {
    issue(d - PD, d, d & 1);   // planes 0 .. PD-1 into their slots (the own-block loads of planes -PD .. -1 are out of range)
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
#pragma unroll
    for (int q = 0; q < NR; q++) raw_bstore<NA * EB>(z, rsG, DW_OOB, 0);
  }

  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      if (t >= g.T) break;
      const int cur = d & 1, prv = cur ^ 1;                        // compile-time after unrolling (t0 % UN == 0)
      const int pm1 = (d + 2) % 3, p0 = d % 3, pp1 = (d + 1) % 3;  // accumulators of planes t-1, t, t+1
      __syncthreads();
      stage(d % PD);
      __syncthreads();
      issue(t, d % PD, cur);
      if (active && !x_nomath) {
        float winB[2][BW];
#pragma unroll
        for (int kh = 0; kh < 3; kh++) lds_window<WIN, 4>(Al + (r * S + kh) * g.LP + wo0 * S, winA2[cur][kh]);
#pragma unroll
        for (int q = 0; q < 2; q++) lds_window<BW, 2>(Bl + (r + q) * a.LPB + wo0, winB[q]);
#pragma unroll
        for (int i = 0; i < SW; i++) dBs[cur][i] = winB[1][i + 1];
        // weight gradient: planes (t, t), (t-1 of dB, t), (t, t-1 of the window); all-VGPR FMAs
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++) {
#pragma unroll
            for (int i = 0; i < SW; i++) {
              dW[9 + kh * 3 + kw] += dBs[cur][i] * winA2[cur][kh][i * S + kw];
              dW[18 + kh * 3 + kw] += dBs[prv][i] * winA2[cur][kh][i * S + kw];
              dW[kh * 3 + kw] += dBs[cur][i] * winA2[prv][kh][i * S + kw];
            }
          }
        // data gradient: output (r, i) reaches the 2 x 2 input quads of dB rows r - 1, r / columns i - 1, i.  The quad's four
        // parities:  even row, even col: taps (0,0) b11, (0,2) b10, (2,0) b01, (2,2) b00;  even row, odd col: (0,1) b11, (2,1) b01;
        // odd row, even col: (1,0) b11, (1,2) b10;  odd row, odd col: (1,1) b11.  kt = 2 opens plane t + 1: products, not sums.
#pragma unroll
        for (int i = 0; i < SW; i++) {
          const float b11 = winB[1][i + 1], b10 = winB[1][i], b01 = winB[0][i + 1], b00 = winB[0][i];
#pragma unroll
          for (int kt = 0; kt < 3; kt++) {
            const float* wk = &wgt[kt * 9];
            const int p = kt == 0 ? pm1 : (kt == 1 ? p0 : pp1);
            v2f e0, e1;
            if (kt == 2) { e0 = (v2f){wk[0], wk[1]} * bc2(b11); e1 = (v2f){wk[3], wk[4]} * bc2(b11); }
            else {
              e0 = pk_fma((v2f){wk[0], wk[1]}, bc2(b11), dAr[p][0][i]);
              e1 = pk_fma((v2f){wk[3], wk[4]}, bc2(b11), dAr[p][1][i]);
            }
            e0 = pk_fma((v2f){wk[6], wk[7]}, bc2(b01), e0);
            e0.x = __builtin_fmaf(wk[2], b10, e0.x);
            e0.x = __builtin_fmaf(wk[8], b00, e0.x);
            e1.x = __builtin_fmaf(wk[5], b10, e1.x);
            dAr[p][0][i] = e0; dAr[p][1][i] = e1;
          }
        }
      }
      emit(t - 1, t >= 1, dAr[pm1], winA2[prv], ownr[prv]);   // plane t-1 is complete now; its accumulator becomes plane t+2's
    }
  }
  // after the loop: plane T-1 waits in dAr[(T-1) % 3], its window in winA2[(T-1) & 1], its araw block in ownr[(T-1) & 1]
  // (constant indices in every branch: a select between two array elements would put the arrays into scratch memory)
  {
    const int m1 = (g.T + 2) % 3, w1 = (g.T + 1) & 1;
    if (w1 == 0) {
      if (m1 == 0) emit(g.T - 1, true, dAr[0], winA2[0], ownr[0]);
      else if (m1 == 1) emit(g.T - 1, true, dAr[1], winA2[0], ownr[0]);
      else emit(g.T - 1, true, dAr[2], winA2[0], ownr[0]);
    } else {
      if (m1 == 0) emit(g.T - 1, true, dAr[0], winA2[1], ownr[1]);
      else if (m1 == 1) emit(g.T - 1, true, dAr[1], winA2[1], ownr[1]);
      else emit(g.T - 1, true, dAr[2], winA2[1], ownr[1]);
    }
  }

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const float s1 = s1p.x + s1p.y, s2 = s2p.x + s2p.y;
  float red[29];
#pragma unroll
  for (int k = 0; k < 29; k++) red[k] = wave_sum_lane63(k < 27 ? dW[k] : (k == 27 ? s1 : s2));
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 29; k++) scratch[k * 4 + wid] = red[k];
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    float v = 0.f;
    for (int w = 0; w < nw; w++) v += scratch[threadIdx.x * 4 + w];
    if (threadIdx.x < 27) atomicAdd(&a.dw[c * 27 + threadIdx.x], v);
    else atomic_add_d(&a.a_sums[c * 2 + (threadIdx.x - 27)], (double)v);
  }
}

// ---- the same kernel on an LDS RING: three activation planes, two dB planes, ONE barrier per plane, FOUR waves per SIMD ----
// The activation window is read ONE PLANE LATE: iteration t stages plane t (nobody reads that buffer any more: its last reader
// was iteration t - 2, a barrier ago) and works on the window of plane t - 1 -- whose weight-gradient taps pair it with
// dB(t - 2), dB(t - 1) and dB(t), two registers each, instead of keeping a second 15-register window alive across iterations; it
// is also the window whose ReLU mask the emit of plane t - 1 needs.  111 VGPRs: four workgroups per CU (35 KB of LDS each).
// Idle threads (rows past the tile) read their windows from zero rows kept behind every plane buffer, so the tap sums need no
// branch.  After the last plane one drain step pairs window T - 1 with dB(T - 1), dB(T - 2) and emits plane T - 1.
#ifndef S2R_OCC
#define S2R_OCC 4
#endif
#ifndef S2R_PD
#define S2R_PD 2
#endif
#define DW_S2_SA 2432   // floats per activation buffer (>= (RIN + 3) * LP: the plane and three zero rows behind it; host check)
#define DW_S2_SB 704    // floats per dB buffer (>= (RB + 2) * LPB)
template <typename T, int CV, int PD, int UN>
__global__ __launch_bounds__(256, S2R_OCC) void dw3d_bwd_s2r_kernel(const DwBwdArgs a) {
  constexpr int S = 2, SW = 2, WIN = 5, BW = 3, NA = 4, NR = 2;
  constexpr int VA = CV, VB = CV / 2, EB = (int)sizeof(T);
  static_assert(UN % 6 == 0 && UN % PD == 0, "roles have periods 2 and 3; slots period PD");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  float* const Al = lds;                      // plane p: Al + (p % 3) * DW_S2_SA
  float* const Bl = lds + 3 * DW_S2_SA;       // plane p: Bl + (p & 1) * DW_S2_SB
  float* const scratch = Bl + 2 * DW_S2_SB;

  int b = blockIdx.x;
  const int tile = __builtin_amdgcn_readfirstlane(b % g.ntile_h); b /= g.ntile_h;
  const int c = __builtin_amdgcn_readfirstlane(b % g.C);
  const int n = __builtin_amdgcn_readfirstlane(b / g.C);
  const int h0 = tile * g.TH;
  const int th_here = min(g.TH, g.Ho - h0);
  const int r = threadIdx.x / g.nstrips, sidx = threadIdx.x - r * g.nstrips;
  const bool active = r < th_here;
  const int ho = h0 + r, wo0 = sidx * SW;

  for (int i = threadIdx.x; i < 3 * DW_S2_SA + 2 * DW_S2_SB; i += blockDim.x) lds[i] = 0.f;

  float wgt[27];
#pragma unroll
  for (int k = 0; k < 27; k++) wgt[k] = a.w[c * 27 + k];
  const float sc = a.ss_a[c * 2], sh = a.ss_a[c * 2 + 1];
  const float* cf = a.coef_nc + ((long long)n * g.C + c) * 4;
  const float cA = cf[0], cB = cf[1], cC = cf[2];

  const int iplB = g.H * g.W * EB, oplB = g.Ho * g.Wo * EB;
  const long long chan = (long long)n * g.C + c;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.araw + chan * g.T * g.H * g.W, 0, g.T * iplB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.ga + chan * g.T * g.H * g.W, 0, g.T * iplB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.dv + chan * g.T * g.Ho * g.Wo, 0, g.T * oplB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.braw + chan * g.T * g.Ho * g.Wo, 0, g.T * oplB, 0x00020000);
  const int rowA0 = h0 * S - g.ph;

  int gA = DW_OOB, lA = 0, gB = DW_OOB, lB = 0;
  bool okA = false, okB = false;
  {
    const int nvr = g.W / VA, v = threadIdx.x;
    if (v < g.RIN * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = rowA0 + lr;
      if (hi >= 0 && hi < g.H) { okA = true; gA = (hi * g.W + jv * VA) * EB; lA = lr * g.LP + jv * VA; }
    }
  }
  {  // dB plane: lds row 0 <-> output row h0-1, col 0 <-> col -1
    const int nvr = g.Wo / VB, v = threadIdx.x;
    if (v < a.RB * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = h0 - 1 + lr;
      if (hi >= 0 && hi < g.Ho) { okB = true; gB = (hi * g.Wo + jv * VB) * EB; lB = lr * a.LPB + 1 + jv * VB; }
    }
  }
  const int hA = ho * 2 - g.ph, wA0 = wo0 * 2;
  int oOwn[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int h = hA + q;
    oOwn[q] = (active && h >= 0 && h < g.H) ? (h * g.W + wA0) * EB : DW_OOB;
  }
#ifdef X3D_EXPERIMENTS
  const bool x_nomath = a.exp & 1, x_noload = a.exp & 2, x_noemit = a.exp & 4;
  if (x_noload) { gA = DW_OOB; gB = DW_OOB; oOwn[0] = DW_OOB; oOwn[1] = DW_OOB; }
#else
  constexpr bool x_nomath = false, x_noload = false, x_noemit = false;
#endif

  Raw sA[PD], sD[PD], sR[PD];   // slot p % PD: the staging vectors (araw, dv, b_raw) of plane p
  Raw ownr[2][NR];              // [p & 1]: the araw block of plane p (loaded in iteration p for the emit of iteration p + 1)
  auto issue_adr = [&](int t, int sl) {   // plane offsets in the scalar offset (range-checked like the vector one)
    raw_bload<VA * EB>(sA[sl], rsA, gA, t * iplB);
    raw_bload<VB * EB>(sD[sl], rsD, gB, t * oplB);
    raw_bload<VB * EB>(sR[sl], rsR, gB, t * oplB);
  };
  auto issue_own = [&](int t, int par) {
#pragma unroll
    for (int q = 0; q < NR; q++) raw_bload<NA * EB>(ownr[par][q], rsA, oOwn[q], t * iplB);
  };

  v2f dAr[3][NR][SW];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int q = 0; q < NR; q++)
#pragma unroll
      for (int i = 0; i < SW; i++) dAr[k][q][i] = (v2f){0.f, 0.f};
  float dW[27];
#pragma unroll
  for (int k = 0; k < 27; k++) dW[k] = 0.f;
  float dBs[3][SW];   // [p % 3]: the own dB strip of plane p
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int i = 0; i < SW; i++) dBs[k][i] = 0.f;
  v2f s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int q = 0; q < NR; q++) ownr[k][q].w[0] = ownr[k][q].w[1] = ownr[k][q].w[2] = ownr[k][q].w[3] = 0u;

  auto emit = [&](int t, bool live, const v2f (&v)[NR][SW], const float (&wm)[3][WIN], const Raw (&own)[NR]) {
#pragma unroll
    for (int q = 0; q < NR; q++) {
      Raw o;
#pragma unroll
      for (int j = 0; j < SW; j++) {
        v2f gp, ap = {raw_get<T>(own[q], 2 * j), raw_get<T>(own[q], 2 * j + 1)};
        if (x_noemit) gp = v[q][j];
        else {
          gp.x = wm[q][2 * j] > 0.f ? v[q][j].x : 0.f;
          gp.y = wm[q][2 * j + 1] > 0.f ? v[q][j].y : 0.f;
          s1p += gp;
          s2p = pk_fma(gp, ap, s2p);
        }
        o.w[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(gp, typename HV<T>::x2));
      }
      raw_bstore<NA * EB>(o, rsG, oOwn[q], (live && !x_noload) ? t * iplB : DW_OOB);
    }
  };
  const v2f sc2 = bc2(sc), sh2 = bc2(sh), cA2 = bc2(cA), cB2 = bc2(cB), cC2 = bc2(cC);
  auto stage = [&](int sl, float* Ab, float* Bb) {
    if (okA) {
      float* dst = Ab + lA;
#pragma unroll
      for (int e = 0; e < VA; e += 2) {
        v2f z = pk_fma((v2f){raw_get<T>(sA[sl], e), raw_get<T>(sA[sl], e + 1)}, sc2, sh2);
        z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f);
        *(v2f*)(dst + e) = z;
      }
    }
    if (okB) {
      float* dst = Bb + lB;
#pragma unroll
      for (int e = 0; e < VB; e += 2) {
        const v2f u = pk_fma((v2f){raw_get<T>(sR[sl], e), raw_get<T>(sR[sl], e + 1)}, cB2, cC2);
        const v2f z = pk_fma((v2f){raw_get<T>(sD[sl], e), raw_get<T>(sD[sl], e + 1)}, cA2, u);
        dst[e] = z.x; dst[e + 1] = z.y;
      }
    }
  };
  // idle threads (rows past the tile) read their windows from the zero rows behind the planes: no branch around the tap sums,
  // zero products, an all-false mask
  const int awin = active ? r * S * g.LP + wo0 * S : g.RIN * g.LP;
  const int bwin = active ? r * a.LPB + wo0 : a.RB * a.LPB;
  auto window_a = [&](const float* Ab, float (&winA)[3][WIN]) {
#pragma unroll
    for (int kh = 0; kh < 3; kh++) lds_window<WIN, 4>(Ab + awin + kh * g.LP, winA[kh]);
  };
  // the weight-gradient taps of activation plane p (its window) against dB(p + 1), dB(p), dB(p - 1) = taps kt = 0, 1, 2
  auto wgrad = [&](const float (&winA)[3][WIN], int bn, int b0, int bp, bool with_next) {
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
      for (int kw = 0; kw < 3; kw++) {
#pragma unroll
        for (int i = 0; i < SW; i++) {
          if (with_next) dW[kh * 3 + kw] += dBs[bn][i] * winA[kh][i * S + kw];
          dW[9 + kh * 3 + kw] += dBs[b0][i] * winA[kh][i * S + kw];
          dW[18 + kh * 3 + kw] += dBs[bp][i] * winA[kh][i * S + kw];
        }
      }
  };

  // prologue: the same load / store sequence as a steady-state iteration (stores dropped)
#pragma unroll
  for (int d = 0; d < PD; d++) {
    issue_adr(d, d);
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
#pragma unroll
    for (int q = 0; q < NR; q++) raw_bstore<NA * EB>(z, rsG, DW_OOB, 0);
  }
  __syncthreads();   // the zero fill
  int dlast = 0;     // the iteration slot (t mod 6) the loop stopped at
  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      dlast = d;
      if (t >= g.T) break;
      const int pm1 = (d + 2) % 3, p0 = d % 3, pp1 = (d + 1) % 3;   // planes t-1, t, t+1: accumulators, LDS buffers, dB strips
      stage(d % PD, Al + p0 * DW_S2_SA, Bl + (d & 1) * DW_S2_SB);
      issue_adr(t + PD, d % PD);
      issue_own(t, d & 1);
      __syncthreads();
      float winA[3][WIN];
#ifdef X3D_EXPERIMENTS
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int j = 0; j < WIN; j++) winA[kh][j] = 0.f;
#endif
      if (!x_nomath) {
        float winB[2][BW];
        window_a(Al + pm1 * DW_S2_SA, winA);   // plane t-1 (t = 0: the zero fill)
#pragma unroll
        for (int q = 0; q < 2; q++) lds_window<BW, 2>(Bl + (d & 1) * DW_S2_SB + bwin + q * a.LPB, winB[q]);
#pragma unroll
        for (int i = 0; i < SW; i++) dBs[p0][i] = winB[1][i + 1];
        wgrad(winA, p0, pm1, pp1, true);        // dB(t), dB(t-1), dB(t-2) (the strip of plane t+1's slot is still plane t-2's)
        // data gradient: output (r, i) reaches the 2 x 2 input quads of dB rows r - 1, r / columns i - 1, i.  The quad's four
        // parities:  even row, even col: taps (0,0) b11, (0,2) b10, (2,0) b01, (2,2) b00;  even row, odd col: (0,1) b11, (2,1) b01;
        // odd row, even col: (1,0) b11, (1,2) b10;  odd row, odd col: (1,1) b11
#pragma unroll
        for (int i = 0; i < SW; i++) {
          const float b11 = winB[1][i + 1], b10 = winB[1][i], b01 = winB[0][i + 1], b00 = winB[0][i];
#pragma unroll
          for (int kt = 0; kt < 3; kt++) {
            const float* wk = &wgt[kt * 9];
            const int p = kt == 0 ? pm1 : (kt == 1 ? p0 : pp1);
            v2f e0, e1;
            if (kt == 2) { e0 = (v2f){wk[0], wk[1]} * bc2(b11); e1 = (v2f){wk[3], wk[4]} * bc2(b11); }   // opens plane t + 1
            else {
              e0 = pk_fma((v2f){wk[0], wk[1]}, bc2(b11), dAr[p][0][i]);
              e1 = pk_fma((v2f){wk[3], wk[4]}, bc2(b11), dAr[p][1][i]);
            }
            e0 = pk_fma((v2f){wk[6], wk[7]}, bc2(b01), e0);
            e0.x = __builtin_fmaf(wk[2], b10, e0.x);
            e0.x = __builtin_fmaf(wk[8], b00, e0.x);
            e1.x = __builtin_fmaf(wk[5], b10, e1.x);
            dAr[p][0][i] = e0; dAr[p][1][i] = e1;
          }
        }
      }
      emit(t - 1, t >= 1, dAr[pm1], winA, ownr[(d + 1) & 1]);   // plane t-1 is complete; winA is its window (idle threads: zeros)
      if (d == UN - 1) dlast = 0;   // a whole pass: the next one (or the drain) starts at slot 0
    }
  }
  // drain: plane T-1's window against dB(T-1), dB(T-2) (there is no dB(T)), then its emit.  T mod 6 fixes every role.
  auto drain = [&](int d) {   // d = T mod 6, a constant at each call
    const int pm1 = (d + 2) % 3, pp1 = (d + 1) % 3;
    float winA[3][WIN];
#ifdef X3D_EXPERIMENTS
#pragma unroll
    for (int kh = 0; kh < 3; kh++)
#pragma unroll
      for (int j = 0; j < WIN; j++) winA[kh][j] = 0.f;
#endif
    if (!x_nomath) {
      window_a(Al + pm1 * DW_S2_SA, winA);
      wgrad(winA, 0, pm1, pp1, false);
    }
    emit(g.T - 1, true, dAr[pm1], winA, ownr[(d + 1) & 1]);
  };
  switch (dlast) {
    case 0: drain(0); break;
    case 1: drain(1); break;
    case 2: drain(2); break;
    case 3: drain(3); break;
    case 4: drain(4); break;
    default: drain(5); break;
  }

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const float s1 = s1p.x + s1p.y, s2 = s2p.x + s2p.y;
  float red[29];
#pragma unroll
  for (int k = 0; k < 29; k++) red[k] = wave_sum_lane63(k < 27 ? dW[k] : (k == 27 ? s1 : s2));
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 29; k++) scratch[k * 4 + wid] = red[k];
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    float v = 0.f;
    for (int w = 0; w < nw; w++) v += scratch[threadIdx.x * 4 + w];
    if (threadIdx.x < 27) atomicAdd(&a.dw[c * 27 + threadIdx.x], v);
    else atomic_add_d(&a.a_sums[c * 2 + (threadIdx.x - 27)], (double)v);
  }
}

// S == 2, strips of two outputs, aligned staging vectors of 16 bytes (CV = 8 / 4 elements), pw == 0 (the caller's own_ok)
bool dw_bwd_s2_launch(const DwBwdArgs& a, int dtype, int SW, int cv, int pd, unsigned grid, int bd, size_t lds, hipStream_t st) {
  if (SW != 2 || pd != 2 || a.g.pw != 0 || x3d_env_int("X3D_DW_S2", 1) == 0) return false;   // X3D_DW_S2=0: A/B hook
  if (dtype == X3D_F32 || cv != 8) return false;   // fp32 storage: 168 VGPRs + 68 B of scratch here -- stays on dw3d_bwd_pd_kernel
  // what both kernels below assume (today's call site in dw_bwd.hip guarantees it; a changed call site must fall back, not
  // compute wrong results): one staging vector per thread for either plane, four waves (the scratch of the final sums),
  // 8-byte dB vectors, 32-bit buffer offsets within a channel
  const DwGeom& g = a.g;
  if (bd != 256 || a.vecB != 4 || (g.W % 8) != 0 || (g.Wo % 4) != 0) return false;
  if (g.RIN * (g.W / 8) > bd || a.RB * (g.Wo / 4) > bd) return false;
  if ((long long)g.T * g.H * g.W * 2 >= (1ll << 30)) return false;
  // two LDS plane buffers, one barrier per plane, when the planes fit the constant buffer stride.  X3D_DW_S2_DB=0: A/B hook
  const int ring = x3d_env_int("X3D_DW_S2_RING", 1);   // A/B hook: 0 = the two-barrier kernel
  if (ring && (a.g.RIN + 3) * a.g.LP <= DW_S2_SA && (a.RB + 2) * a.LPB <= DW_S2_SB) {
    if (x3d_describe.out) {
      snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_s2r_kernel<%s, 8, %d, 6>", dtype == X3D_BF16 ? TypeName<bf16>::v : TypeName<f16>::v, S2R_PD);
      return true;
    }
    const size_t ldsr = (3 * DW_S2_SA + 2 * DW_S2_SB + 29 * 4 + 8) * sizeof(float);
    if (dtype == X3D_BF16) hipLaunchKernelGGL((dw3d_bwd_s2r_kernel<bf16, 8, S2R_PD, 6>), dim3(grid), dim3(bd), ldsr, st, a);
    else hipLaunchKernelGGL((dw3d_bwd_s2r_kernel<f16, 8, S2R_PD, 6>), dim3(grid), dim3(bd), ldsr, st, a);
    return true;
  }
  if (x3d_describe.out) {
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_s2_kernel<%s, 8, 2, 6>", dtype == X3D_BF16 ? TypeName<bf16>::v : TypeName<f16>::v);
    return true;
  }
  if (dtype == X3D_BF16) hipLaunchKernelGGL((dw3d_bwd_s2_kernel<bf16, 8, 2, 6>), dim3(grid), dim3(bd), lds, st, a);
  else hipLaunchKernelGGL((dw3d_bwd_s2_kernel<f16, 8, 2, 6>), dim3(grid), dim3(bd), lds, st, a);
  return true;
}
