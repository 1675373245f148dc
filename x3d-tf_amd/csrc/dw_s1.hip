// Fused backward of the channelwise 3x3x3 convolution, STRIDE 1, strips of four outputs, rows that are whole aligned 16-byte
// staging vectors in 16-bit storage (X3D-M stage 2: 54 channels at 56 x 56, two launches of 477 us per train step; X3D-S: 80 x 80,
// 40 x 40).  The form of dw3d_bwd_s2r_kernel (dw_s2.hip), which see: the one-plane-ahead kernel it replaces
// (dw3d_bwd_kernel<T, 1, 4, 1, 8>, dw_bwd.hip) issues 354 vector instructions per thread and plane for 216 multiply-adds --
// 34 register copies (window -> previous window, accumulator rotation), ~25 of 64-bit address arithmetic for its global
// loads and stores, the ReLU mask recomputed in the emit.  Here:
//   * register ROLES (T loop unrolled by 6), accumulators opened by a product instead of zeroed;
//   * buffer instructions with the plane offset in the SCALAR offset: per-thread offsets are loop constants;
//   * an LDS RING of three activation planes and two dB planes, ONE barrier per plane: the activation window is read one plane
//     late (iteration t works on the window of plane t - 1 against dB(t), dB(t - 1), dB(t - 2) -- four registers each -- so only
//     one 18-register window is alive), and it is the window whose centre row carries the ReLU mask of the strip the thread
//     emits (row ho, columns wo0 .. wo0 + 3 = window row 1, columns 1 .. 4: TF-SAME pads of 1);
//   * idle threads read their windows from zero rows behind every plane buffer: no branch around the tap sums;
//   * ONE plane in flight per workgroup in a register slot (as in the kernel it replaces), the araw strip for the sum of
//     ga * araw loaded one iteration ahead of its emit: 102 VGPRs, FOUR waves per SIMD (31 KB of LDS per workgroup).
// Measured (64 clips, 54 channels, 16 x 56 x 56, bf16; profiles/r05_ab_dw_s1.txt): 289 vector instructions per thread and plane
// instead of 354, and yet the isolated launch does not move -- 568-574 us before, 575-581 with one plane in flight at four waves,
// 589-599 with two or three planes in flight at three waves (132 / 144 VGPRs); with every global access out of range it keeps
// 497 us, with the tap sums skipped 353 -- fewer instructions did not buy time, so issue slots are not what this layer waits
// for (unexplained).  Inside the train step the four-wave form does measure faster, and that is why it is the one shipped:
// 497-503 -> 470-472 us per launch on one box, alternating runs (two launches per step).
// Sums are taken in a different order than in dw3d_bwd_kernel (tests: fp64 reference of the rounded inputs).
#include "dw_common.h"

#ifndef S1R_OCC
#define S1R_OCC 4
#endif
#ifndef S1R_PD
#define S1R_PD 1
#endif
#define DW_S1_SA 1536   // floats per activation buffer (>= (RIN + 3) * LP: the plane and three zero rows behind it; host check)
#define DW_S1_SB 1536   // floats per dB buffer (>= (RB + 3) * LPB)
template <typename T, int CV, int PD, int UN>
__global__ __launch_bounds__(256, S1R_OCC) void dw3d_bwd_s1r_kernel(const DwBwdArgs a) {
  constexpr int SW = 4, WIN = 6, BW = 6;
  constexpr int VA = CV, VB = CV, EB = (int)sizeof(T);
  static_assert(UN % 6 == 0 && UN % PD == 0, "roles have periods 2 and 3; slots period PD");
  static_assert(VA % 2 == 0 && sizeof(T) == 2, "staging in pairs, 16-bit storage");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  float* const Al = lds;                      // plane p: Al + (p % 3) * DW_S1_SA
  float* const Bl = lds + 3 * DW_S1_SA;       // plane p: Bl + (p & 1) * DW_S1_SB
  float* const scratch = Bl + 2 * DW_S1_SB;

  int b = blockIdx.x;
  const int tile = __builtin_amdgcn_readfirstlane(b % g.ntile_h); b /= g.ntile_h;
  const int c = __builtin_amdgcn_readfirstlane(b % g.C);
  const int n = __builtin_amdgcn_readfirstlane(b / g.C);
  const int h0 = tile * g.TH;
  const int th_here = min(g.TH, g.Ho - h0);
  const int r = threadIdx.x / g.nstrips, sidx = threadIdx.x - r * g.nstrips;
  const bool active = r < th_here;
  const int ho = h0 + r, wo0 = sidx * SW;

  for (int i = threadIdx.x; i < 3 * DW_S1_SA + 2 * DW_S1_SB; i += blockDim.x) lds[i] = 0.f;

  float wgt[27];
#pragma unroll
  for (int k = 0; k < 27; k++) wgt[k] = a.w[c * 27 + k];
  const float sc = a.ss_a[c * 2], sh = a.ss_a[c * 2 + 1];
  const float* cf = a.coef_nc + ((long long)n * g.C + c) * 4;
  const float cA = cf[0], cB = cf[1], cC = cf[2];

  const int plB = g.H * g.W * EB;   // stride 1: input and output planes have one size
  const long long chan = (long long)n * g.C + c;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.araw + chan * g.T * g.H * g.W, 0, g.T * plB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.ga + chan * g.T * g.H * g.W, 0, g.T * plB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.dv + chan * g.T * g.H * g.W, 0, g.T * plB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.braw + chan * g.T * g.H * g.W, 0, g.T * plB, 0x00020000);

  // staging maps: ONE vector per thread and tensor (host guarantees it).  Both planes: LDS row 0 <-> image row h0 - 1,
  // LDS column 0 <-> image column -1 (the TF-SAME pads of a 3-tap stride-1 convolution)
  int gA = DW_OOB, lA = 0, gB = DW_OOB, lB = 0;
  bool okA = false, okB = false;
  {
    const int nvr = g.W / VA, v = threadIdx.x;
    if (v < g.RIN * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = h0 - 1 + lr;
      if (hi >= 0 && hi < g.H) { okA = true; gA = (hi * g.W + jv * VA) * EB; lA = lr * g.LP + 1 + jv * VA; }
    }
    if (v < a.RB * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = h0 - 1 + lr;
      if (hi >= 0 && hi < g.H) { okB = true; gB = (hi * g.W + jv * VB) * EB; lB = lr * a.LPB + 1 + jv * VB; }
    }
  }
  // the strip of dA this thread owns: row ho, columns wo0 .. wo0 + 3 = window row 1, columns 1 .. 4
  const int oOwn = active ? (ho * g.W + wo0) * EB : DW_OOB;
#ifdef X3D_EXPERIMENTS
  const bool x_nomath = a.exp & 1, x_noload = a.exp & 2, x_noemit = a.exp & 4;
  int gA_ = gA, gB_ = gB, oOwn_ = oOwn;
  if (x_noload) { gA_ = DW_OOB; gB_ = DW_OOB; oOwn_ = DW_OOB; }
#else
  constexpr bool x_nomath = false, x_noload = false, x_noemit = false;
  const int gA_ = gA, gB_ = gB, oOwn_ = oOwn;
#endif

  Raw sA[PD], sD[PD], sR[PD];   // slot p % PD: the staging vectors (araw, dv, b_raw) of plane p
  Raw ownr[2];                  // [p & 1]: the araw strip of plane p (loaded in iteration p for the emit of iteration p + 1)
  auto issue_adr = [&](int t, int sl) {   // plane offsets in the scalar offset (range-checked like the vector one)
    raw_bload<VA * EB>(sA[sl], rsA, gA_, t * plB);
    raw_bload<VB * EB>(sD[sl], rsD, gB_, t * plB);
    raw_bload<VB * EB>(sR[sl], rsR, gB_, t * plB);
  };
  auto issue_own = [&](int t, int par) { raw_bload<SW * EB>(ownr[par], rsA, oOwn_, t * plB); };

  float dAr[3][SW];     // dAr[p % 3] = gradient plane p while it is being accumulated (opened by a product: never zeroed)
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int i = 0; i < SW; i++) dAr[k][i] = 0.f;
  float dW[27];
#pragma unroll
  for (int k = 0; k < 27; k++) dW[k] = 0.f;
  float dBs[3][SW];     // [p % 3]: the own dB strip of plane p
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int i = 0; i < SW; i++) dBs[k][i] = 0.f;
  v2f s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; k++) ownr[k].w[0] = ownr[k].w[1] = ownr[k].w[2] = ownr[k].w[3] = 0u;

  // plane p is complete: mask with ReLU'(BN_a(a)) read off the centre row of plane p's window (relu(sc a + sh); zero rows for
  // idle threads and rows outside the image), per-channel sums, store.  !live (p < 0): the window is the zero fill.
  auto emit = [&](int t, bool live, const float (&v)[SW], const float (&wm)[3][WIN], const Raw& own) {
    Raw o;
#pragma unroll
    for (int j = 0; j < SW / 2; j++) {
      v2f gp, ap = {raw_get<T>(own, 2 * j), raw_get<T>(own, 2 * j + 1)};
      if (x_noemit) { gp.x = v[2 * j]; gp.y = v[2 * j + 1]; }
      else {
        gp.x = wm[1][1 + 2 * j] > 0.f ? v[2 * j] : 0.f;
        gp.y = wm[1][2 + 2 * j] > 0.f ? v[2 * j + 1] : 0.f;
        s1p += gp;
        s2p = pk_fma(gp, ap, s2p);
      }
      o.w[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(gp, typename HV<T>::x2));
    }
    raw_bstore<SW * EB>(o, rsG, oOwn_, (live && !x_noload) ? t * plB : DW_OOB);
  };
  const v2f sc2 = bc2(sc), sh2 = bc2(sh), cA2 = bc2(cA), cB2 = bc2(cB), cC2 = bc2(cC);
  auto stage = [&](int sl, float* Ab, float* Bb) {
    if (okA) {
      float* dst = Ab + lA;
#pragma unroll
      for (int e = 0; e < VA; e += 2) {
        v2f z = pk_fma((v2f){raw_get<T>(sA[sl], e), raw_get<T>(sA[sl], e + 1)}, sc2, sh2);
        dst[e] = fmaxf(z.x, 0.f); dst[e + 1] = fmaxf(z.y, 0.f);
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
  // idle threads (rows past the tile) read their windows from the zero rows behind the planes
  const int awin = active ? r * g.LP + wo0 : g.RIN * g.LP;
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
          if (with_next) dW[kh * 3 + kw] += dBs[bn][i] * winA[kh][i + kw];
          dW[9 + kh * 3 + kw] += dBs[b0][i] * winA[kh][i + kw];
          dW[18 + kh * 3 + kw] += dBs[bp][i] * winA[kh][i + kw];
        }
      }
  };

  // prologue: the same load / store sequence as a steady-state iteration (stores dropped)
#pragma unroll
  for (int d = 0; d < PD; d++) {
    issue_adr(d, d);
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
    raw_bstore<SW * EB>(z, rsG, DW_OOB, 0);
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
      stage(d % PD, Al + p0 * DW_S1_SA, Bl + (d & 1) * DW_S1_SB);
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
        float winB[3][BW];
        window_a(Al + pm1 * DW_S1_SA, winA);   // plane t-1 (t = 0: the zero fill)
#pragma unroll
        for (int q = 0; q < 3; q++) lds_window<BW, 4>(Bl + (d & 1) * DW_S1_SB + bwin + q * a.LPB, winB[q]);
#pragma unroll
        for (int i = 0; i < SW; i++) dBs[p0][i] = winB[1][i + 1];
        wgrad(winA, p0, pm1, pp1, true);        // dB(t), dB(t-1), dB(t-2) (the strip in plane t+1's place is still plane t-2's)
        // data gradient: dA[t + kt - 1][ho][wo0 + i] += w[kt][kh][kw] * dB[t][ho + 1 - kh][wo0 + i + 1 - kw]; kt = 2 opens plane t + 1
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int i = 0; i < SW; i++) {
              const float v = winB[2 - kh][i + 2 - kw];
              dAr[pm1][i] += wgt[kh * 3 + kw] * v;
              dAr[p0][i] += wgt[9 + kh * 3 + kw] * v;
              if (kh == 0 && kw == 0) dAr[pp1][i] = wgt[18] * v;
              else dAr[pp1][i] += wgt[18 + kh * 3 + kw] * v;
            }
      }
      emit(t - 1, t >= 1, dAr[pm1], winA, ownr[(d + 1) & 1]);   // plane t-1 is complete; winA is its window
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
      window_a(Al + pm1 * DW_S1_SA, winA);
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

// S == 1, strips of four outputs, rows of whole 16-byte vectors in 16-bit storage, one staging vector per thread and tensor
bool dw_bwd_s1_launch(const DwBwdArgs& a, int dtype, int SW, int cv, unsigned grid, int bd, hipStream_t st) {
  if (SW != 4 || cv != 8 || dtype == X3D_F32 || bd != 256 || x3d_env_int("X3D_DW_S1R", 1) == 0) return false;   // X3D_DW_S1R=0: A/B hook
  const DwGeom& g = a.g;
  if (g.W % 8 || g.nstrips * 4 != g.W || a.vecB != 8) return false;
  if (g.RIN * (g.W / 8) > 256 || a.RB * (g.W / 8) > 256) return false;
  if ((g.RIN + 3) * g.LP > DW_S1_SA || (a.RB + 3) * a.LPB > DW_S1_SB) return false;
  if ((long long)g.T * g.H * g.W * 2 >= (1ll << 30)) return false;   // 32-bit buffer offsets
  if (x3d_describe.out) {
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_s1r_kernel<%s, 8, %d, 6>", dtype == X3D_BF16 ? TypeName<bf16>::v : TypeName<f16>::v, S1R_PD);
    return true;
  }
  const size_t lds = (3 * DW_S1_SA + 2 * DW_S1_SB + 29 * 4 + 8) * sizeof(float);
  if (dtype == X3D_BF16) hipLaunchKernelGGL((dw3d_bwd_s1r_kernel<bf16, 8, S1R_PD, 6>), dim3(grid), dim3(bd), lds, st, a);
  else hipLaunchKernelGGL((dw3d_bwd_s1r_kernel<f16, 8, S1R_PD, 6>), dim3(grid), dim3(bd), lds, st, a);
  return true;
}
