// x3d_dw3d_bwd: channelwise 3x3x3 convolution, fused data + weight gradient (design notes in dw_common.h)
#include "dw_common.h"

// ================================================================================================
// backward (data + weight fused).  Thread = output position strip, exactly as in the forward.
//   dB = A*dv + B*braw + C            (BN_b / SE backward folded, per (n,c) coefficients)
//   dW[kt][kh][kw] += sum dB[t][ho][wo] * act[t+kt-1][ho*S+kh-ph][wo*S+kw-pw]
//   dA[t][h][w]     = sum w[kt][kh][kw] * dB[t+1-kt][(h+ph-kh)/S][(w+pw-kw)/S]
//   ga = dA * [sc*araw + sh > 0];   a_sums += (sum ga, sum ga*araw)
// Planes of act and dB are streamed through LDS once (prefetched one plane ahead); the temporal taps are
// handled with rotating register accumulators (dA) and a one-plane-old register window (dW).
// ================================================================================================

template <typename T, int S, int SW, int NSV, int CV>
__global__ __launch_bounds__(256) void dw3d_bwd_kernel(const DwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  constexpr int WIN = (SW - 1) * S + 3;          // act window columns
  constexpr int BW = (S == 1) ? SW + 2 : SW + 1; // dB window columns
  constexpr int BR = (S == 1) ? 3 : 2;           // dB window rows
  constexpr int NA = (S == 1) ? SW : 2 * SW;     // dA columns owned per row
  constexpr int NR = (S == 1) ? 1 : 2;           // dA rows owned
  constexpr int NS = NSV > 0 ? NSV : 1;
  constexpr bool RAG = CV < 0;                   // ragged planes: flat staging in vectors of -CV elements (FlatMap)
  constexpr int RV = RAG ? -CV : 1;
  // deferred emit (see the plane loop): costs NR*NA registers -- the 2x2-quad stride-2 variant would drop a wave
  constexpr bool DEFER = (S == 1);
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
  auto af = [=](float v) { return fmaxf(sc * v + sh, 0.f); };
  auto bf = [=](float dvv, float bv) { return cA * dvv + cB * bv + cC; };

  const long long ipl = (long long)g.H * g.W, opl = (long long)g.Ho * g.Wo;
  const T* araw = (const T*)a.araw + ((long long)n * g.C + c) * g.T * ipl;
  const T* dvp = (const T*)a.dv + ((long long)n * g.C + c) * g.T * opl;
  const T* brp = (const T*)a.braw + ((long long)n * g.C + c) * g.T * opl;
  T* gap = (T*)a.ga + ((long long)n * g.C + c) * g.T * ipl;
  const int rowA0 = h0 * S - g.ph;
  const int vecA = CV > 0 ? CV : (RAG ? RV : g.vec);
  const int vecB = CV > 0 ? (S == 1 ? CV : (CV > 1 ? CV / 2 : 1)) : (RAG ? RV : a.vecB);

  typename DwSel<RAG, FlatMap<NS>, StageMap<NS>>::type mapA, mapB;
  Raw rawA[NS], rawD[NS], rawR[NS];
  auto issue = [&](int t) {
    if constexpr (NSV > 0) {
#pragma unroll
      for (int i = 0; i < NS; i++) {
        if (mapA.goff[i] >= 0) raw_load<T>(rawA[i], araw + t * ipl + mapA.goff[i], vecA);
        if (mapB.goff[i] >= 0) {
          raw_load<T>(rawD[i], dvp + t * opl + mapB.goff[i], vecB);
          raw_load<T>(rawR[i], brp + t * opl + mapB.goff[i], vecB);
        }
      }
    }
  };
  if constexpr (NSV > 0) {
    mapA.build(g.RIN, g.LP, rowA0, g.H, g.W, g.pw, vecA);
    mapB.build(a.RB, a.LPB, h0 - 1, g.Ho, g.Wo, 1, vecB);   // dB plane: lds row 0 <-> output row h0-1, col 0 <-> col -1
    issue(0);
  }

  float dA0[NR][NA], dA1[NR][NA], dA2[NR][NA], fin[NR][NA];   // fin: finished plane waiting for its deferred emit
#pragma unroll
  for (int q = 0; q < NR; q++)
#pragma unroll
    for (int i = 0; i < NA; i++) { dA0[q][i] = 0.f; dA1[q][i] = 0.f; dA2[q][i] = 0.f; fin[q][i] = 0.f; }
  float dW[27];
#pragma unroll
  for (int k = 0; k < 27; k++) dW[k] = 0.f;
  float winA_prev[3][WIN], dB_prev[SW];
#pragma unroll
  for (int kh = 0; kh < 3; kh++)
#pragma unroll
    for (int j = 0; j < WIN; j++) winA_prev[kh][j] = 0.f;
#pragma unroll
  for (int i = 0; i < SW; i++) dB_prev[i] = 0.f;
  float s1 = 0.f, s2 = 0.f;

  // image rows / cols owned for dA
  const int hA = (S == 1) ? ho : ho * 2 - g.ph;
  const int wA0 = (S == 1) ? wo0 : wo0 * 2 - g.pw;

  // this thread's own araw strip of the plane that is emitted at the end of the iteration: loaded at the
  // top of the iteration so its latency hides behind the plane's arithmetic
  // the NA owned columns of a row are contiguous and NA-aligned when the strips tile the row exactly and the
  // left pad is 0: one vector load / store per row instead of NA two-byte accesses
  // (ragged planes: every strip that lies inside the row, at whatever alignment; the strips cut by a row end go
  // element by element)
  const bool vown = RAG ? (NA > 1 && wA0 >= 0 && wA0 + NA <= g.W) : (CV > 0 && NA > 1) || ((NA > 1) && (g.W % NA == 0) && (S == 1 || g.pw == 0) && (g.Wo % SW == 0) &&
                    (((uintptr_t)a.araw) % (NA * sizeof(T)) == 0) && (((uintptr_t)a.ga) % (NA * sizeof(T)) == 0));
  float ar[NR][NA];
  auto load_own = [&](int t) {
    if (!active) return;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      const int h = hA + q;
      if (vown) {
        if (h >= 0 && h < g.H) VecIO<T, NA>::load(araw + t * ipl + (long long)h * g.W + wA0, ar[q]);
      } else {
#pragma unroll
        for (int i = 0; i < NA; i++) {
          const int w = wA0 + i;
          ar[q][i] = (h >= 0 && h < g.H && w >= 0 && w < g.W) ? to_f<T>(araw[t * ipl + (long long)h * g.W + w]) : 0.f;
        }
      }
    }
  };
  auto emit = [&](int t, const float (&v)[NR][NA]) {
    if (!active) return;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      const int h = hA + q;
      if (h < 0 || h >= g.H) continue;
      const long long base = t * ipl + (long long)h * g.W;
      if (vown) {
        float gv[NA];
#pragma unroll
        for (int i = 0; i < NA; i++) {
          const float av = ar[q][i];
          gv[i] = (sc * av + sh > 0.f) ? v[q][i] : 0.f;
          s1 += gv[i];
          s2 += gv[i] * av;
        }
        VecIO<T, NA>::store(gap + base + wA0, gv);
      } else {
#pragma unroll
        for (int i = 0; i < NA; i++) {
          const int w = wA0 + i;
          if (w >= 0 && w < g.W) {
            const float av = ar[q][i];
            const float gv = (sc * av + sh > 0.f) ? v[q][i] : 0.f;
            gap[base + w] = from_f<T>(gv);
            s1 += gv;
            s2 += gv * av;
          }
        }
      }
    }
  };

  for (int t = 0; t < g.T; ++t) {
    __syncthreads();
    if constexpr (NSV > 0) {
#pragma unroll
      for (int i = 0; i < NS; i++) {
        if (mapA.goff[i] >= 0) {
          float* d = Al + mapA.loff[i];
          if constexpr (RAG) flat_commit<T, RV>(d, mapA.wrap[i], g.LP - g.W, rawA[i], af);
          else {
#pragma unroll
            for (int e = 0; e < MaxVec<T>::v; e++) if (e < vecA) d[e] = af(raw_get<T>(rawA[i], e));
          }
        }
        if (mapB.goff[i] >= 0) {
          float* d = Bl + mapB.loff[i];
          if constexpr (RAG) flat_commit2<T, RV>(d, mapB.wrap[i], a.LPB - g.Wo, rawD[i], rawR[i], bf);
          else {
#pragma unroll
            for (int e = 0; e < MaxVec<T>::v; e++) if (e < vecB) d[e] = bf(raw_get<T>(rawD[i], e), raw_get<T>(rawR[i], e));
          }
        }
      }
    } else {
      stage_direct<T>(araw + t * ipl, Al, g.RIN, g.LP, rowA0, g.H, g.W, g.pw, vecA, af);
      stage_direct2<T>(dvp + t * opl, brp + t * opl, Bl, a.RB, a.LPB, h0 - 1, g.Ho, g.Wo, 1, vecB, bf);
    }
    __syncthreads();
    if (t + 1 < g.T) issue(t + 1);
    // plane t-2 is emitted here, one iteration after it was completed, with the araw strip loaded during the
    // previous iteration: the store and the strip load then have a whole plane of arithmetic to retire before
    // the in-order vmcnt wait for the prefetched planes at the top of the next iteration
    if constexpr (DEFER) { if (t >= 2) emit(t - 2, fin); }
    if (t >= 1) load_own(t - 1);
    if (active) {
      float winA[3][WIN], winB[BR][BW];
#pragma unroll
      for (int kh = 0; kh < 3; kh++) {
        const float* row = Al + (r * S + kh) * g.LP + wo0 * S;
        lds_window<WIN, (SW * S >= 4 ? 4 : SW * S)>(row, winA[kh]);
      }
#pragma unroll
      for (int q = 0; q < BR; q++) {
        const float* row = Bl + (r + q) * a.LPB + wo0;
        lds_window<BW, (SW >= 4 ? 4 : SW)>(row, winB[q]);
      }
      // own dB strip: window row of output row `ho` is index 1 in both layouts; col wo0+i is index i+1
      float dBo[SW];
#pragma unroll
      for (int i = 0; i < SW; i++) dBo[i] = winB[1][i + 1];

      // ---- weight gradient (accumulated straight into dW: a separate partial sum costs one v_add per tap)
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
#pragma unroll
          for (int i = 0; i < SW; i++) {
            dW[9 + kh * 3 + kw] += dBo[i] * winA[kh][i * S + kw];           // kt = 1: dB[t] * act[t]
            dW[18 + kh * 3 + kw] += dB_prev[i] * winA[kh][i * S + kw];      // kt = 2: dB[t-1] * act[t]
            dW[kh * 3 + kw] += dBo[i] * winA_prev[kh][i * S + kw];          // kt = 0: dB[t] * act[t-1]
          }
        }

      // ---- data gradient: plane dB[t] feeds dA[t-1] (kt=0), dA[t] (kt=1), dA[t+1] (kt=2)
      if constexpr (S == 1) {
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int i = 0; i < SW; i++) {
              const float v = winB[2 - kh][i + 2 - kw];   // dB[h+1-kh][w+1-kw]
              dA0[0][i] += wgt[kh * 3 + kw] * v;
              dA1[0][i] += wgt[9 + kh * 3 + kw] * v;
              dA2[0][i] += wgt[18 + kh * 3 + kw] * v;
            }
      } else {
        // rows: q=0 is image row 2ho-ph (taps kh=0 from ho, kh=2 from ho-1); q=1 is 2ho-ph+1 (kh=1 from ho)
        // cols likewise.  winB[1][*] = dB row ho, winB[0][*] = row ho-1; col index i+1 = wo0+i.
#pragma unroll
        for (int i = 0; i < SW; i++) {
          const float b11 = winB[1][i + 1], b10 = winB[1][i], b01 = winB[0][i + 1], b00 = winB[0][i];
#pragma unroll
          for (int kt = 0; kt < 3; kt++) {
            const float* wk = &wgt[kt * 9];
            const float eA = wk[0] * b11 + wk[2] * b10 + wk[6] * b01 + wk[8] * b00;  // (hA, wA)
            const float eB = wk[1] * b11 + wk[7] * b01;                              // (hA, wB)
            const float eC = wk[3] * b11 + wk[5] * b10;                              // (hB, wA)
            const float eD = wk[4] * b11;                                            // (hB, wB)
            if (kt == 0) { dA0[0][2 * i] += eA; dA0[0][2 * i + 1] += eB; dA0[1][2 * i] += eC; dA0[1][2 * i + 1] += eD; }
            if (kt == 1) { dA1[0][2 * i] += eA; dA1[0][2 * i + 1] += eB; dA1[1][2 * i] += eC; dA1[1][2 * i + 1] += eD; }
            if (kt == 2) { dA2[0][2 * i] += eA; dA2[0][2 * i + 1] += eB; dA2[1][2 * i] += eC; dA2[1][2 * i + 1] += eD; }
          }
        }
      }
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int j = 0; j < WIN; j++) winA_prev[kh][j] = winA[kh][j];
#pragma unroll
      for (int i = 0; i < SW; i++) dB_prev[i] = dBo[i];
    }
    if constexpr (!DEFER) {
      // retire every outstanding load (the prefetched planes were issued a whole plane of arithmetic ago) BEFORE the
      // store goes out: the wait at the top of the next iteration then has nothing to wait for, and the store's
      // write latency is never on the critical path (vmcnt retires in order, stores included)
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), lgkmcnt / expcnt untouched
      if (t >= 1) emit(t - 1, dA0);
    }
#pragma unroll
    for (int q = 0; q < NR; q++)
#pragma unroll
      for (int i = 0; i < NA; i++) {
        if constexpr (DEFER) fin[q][i] = dA0[q][i];
        dA0[q][i] = dA1[q][i]; dA1[q][i] = dA2[q][i]; dA2[q][i] = 0.f;
      }
  }
  if constexpr (DEFER) { if (g.T >= 2) emit(g.T - 2, fin); }
  load_own(g.T - 1);
  emit(g.T - 1, dA0);

  // block reduction of the 27 weight-gradient taps and the two BN sums: DPP wave sums that end in lane 63,
  // which alone writes the 29 partials (one predicate around all the LDS writes)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
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

template <typename T, int S, int SW, int CV>
static void dw_bwd_launch_cv(const DwBwdArgs& a, int nsv, unsigned grid, int bd, size_t lds, hipStream_t st) {
  if (x3d_describe.out) {
    const bool gen = nsv > 2;
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_kernel<%s, %d, %d, %d, %d>", TypeName<T>::v,
             S, SW, gen ? 0 : (nsv <= 1 ? 1 : 2), gen ? 0 : CV);
    return;
  }
  if (nsv <= 1) hipLaunchKernelGGL((dw3d_bwd_kernel<T, S, SW, 1, CV>), dim3(grid), dim3(bd), lds, st, a);
  else if (nsv <= 2) hipLaunchKernelGGL((dw3d_bwd_kernel<T, S, SW, 2, CV>), dim3(grid), dim3(bd), lds, st, a);
  else hipLaunchKernelGGL((dw3d_bwd_kernel<T, S, SW, 0, 0>), dim3(grid), dim3(bd), lds, st, a);
}
template <typename T, int S, int SW>
static void dw_bwd_launch_nsv(const DwBwdArgs& a, int nsv, int cv, unsigned grid, int bd, size_t lds, hipStream_t st) {
  switch (cv) {
    case -8: if constexpr (sizeof(T) == 2) { dw_bwd_launch_cv<T, S, SW, -8>(a, nsv, grid, bd, lds, st); break; }
    case -4: dw_bwd_launch_cv<T, S, SW, -4>(a, nsv, grid, bd, lds, st); break;
    case 8: if constexpr (sizeof(T) == 2) { dw_bwd_launch_cv<T, S, SW, 8>(a, nsv, grid, bd, lds, st); break; }
    case 4: dw_bwd_launch_cv<T, S, SW, 4>(a, nsv, grid, bd, lds, st); break;
    case 2: dw_bwd_launch_cv<T, S, SW, 2>(a, nsv, grid, bd, lds, st); break;
    case 1: dw_bwd_launch_cv<T, S, SW, 1>(a, nsv, grid, bd, lds, st); break;
    default: dw_bwd_launch_cv<T, S, SW, 0>(a, nsv, grid, bd, lds, st); break;
  }
}

template <typename T, int S>
static int dw_bwd_launch(const x3d_dw3d_bwd_args* f, hipStream_t st) {
  DwBwdArgs a;
  a.dv = f->dv; a.braw = f->braw; a.coef_nc = f->coef_nc; a.araw = f->araw; a.ss_a = f->a_scale_shift;
  a.w = f->w; a.ga = f->ga; a.a_sums = f->a_sums; a.dw = f->dw;
  const int Wo = ceil_div(f->W, S);
  int SW = dw_pick_sw(Wo);
  if (S == 2 && SW > 2) SW = 2;  // 2x2 input quads per output: keep the register footprint bounded
  int bd; size_t ldsf;
  if (dw_geom(a.g, f->N, f->C, f->T, f->H, f->W, S, SW, sizeof(T), f->araw, f->ga, nullptr, &bd, &ldsf)) {
    x3d_set_error("dw3d_bwd: row of %d outputs does not fit one workgroup", Wo);
    return X3D_ERR_INVALID;
  }
  a.exp = x3d_env_int("X3D_DW_PD_EXP", 0);   // result-changing timing hooks: -DX3D_EXPERIMENTS builds only
  a.RB = (S == 1) ? a.g.TH + 2 : a.g.TH + 1;
  a.LPB = (a.g.nstrips * SW + ((S == 1) ? 2 : 1) + 3) & ~3;   // multiple of 4 floats (aligned window reads)
  a.vecB = pick_vec(sizeof(T), a.g.Wo, f->dv, f->braw);
  const size_t lds = (ldsf + (size_t)a.RB * a.LPB + 29 * 4 + 8) * sizeof(float);
  X3D_REQUIRE(lds <= 64 * 1024, "dw3d_bwd: tile needs %zu B of LDS", lds);
  const long long grid = (long long)f->N * f->C * a.g.ntile_h;
  X3D_REQUIRE(grid < (1ll << 31), "dw3d_bwd: grid too large");
  // compile-time staging widths (cvA along W for araw, cvB = cvA or cvA/2 along Wo for dv/braw) when the
  // owned dA columns form whole aligned vectors
  const int NA = (S == 1) ? SW : 2 * SW;
  const bool own_ok = (a.g.W % NA == 0) && (S == 1 || a.g.pw == 0) && (a.g.Wo % SW == 0) &&
                      (((uintptr_t)f->araw) % (NA * sizeof(T)) == 0) && (((uintptr_t)f->ga) % (NA * sizeof(T)) == 0);
  int cv = own_ok ? a.g.vec : 0;
  if (cv > 0) {
    const int cvB = (S == 1) ? cv : (cv > 1 ? cv / 2 : 1);
    if (a.vecB % cvB != 0) cv = 0;             // dv / braw rows must admit the derived width
    else a.vecB = cvB;
  }
  const int nsvA = dw_nsv(a.g.RIN, a.g.W, a.g.vec, bd), nsvB = dw_nsv(a.RB, a.g.Wo, a.vecB, bd);
  int nsv = nsvA > nsvB ? nsvA : nsvB;
  // ragged rows (39, 78, 91 ... wide; stride 2 with a left pad; a misaligned tensor): flat staging with unaligned
  // 16 / 8-byte vectors and per-thread vector / scalar own strips instead of the unprefetched generic path.
  // X3D_DW_FLAT=0: A/B hook.
  const int flat_env = x3d_env_int("X3D_DW_FLAT", 1);
  if ((cv == 0 || flat_env == 2) && flat_env != 0) {   // 2: force (experiment)
    const int rv = dw_flat_vec(sizeof(T), a.g.W < a.g.Wo ? a.g.W : a.g.Wo);
    if (rv > 0) {
      const int fa = dw_nsv_flat(a.g.RIN, a.g.W, rv, bd), fb = dw_nsv_flat(a.RB, a.g.Wo, rv, bd);
      if (fa <= 2 && fb <= 2) { cv = -rv; nsv = fa > fb ? fa : fb; }
    }
  }
  // small planes (strips of 1 / 2 outputs, stride 1): deep-prefetch variant (dw_pd.hip) when one staging vector per
  // thread and tensor covers the tile
  // stride 2: depth 2 (144 VGPRs, 3 waves; no vmcnt(0) drain before the stores: 1066 -> 961 us at 112x112); depth 4 is
  // 160-190 VGPRs and slower.  X3D_DW_PD_S2=1 / X3D_DW_PD=1 switch back to the one-plane-ahead kernel (A/B hooks).
  const int pd = S == 1 ? dw_pick_pd(SW) : (dw_pick_pd(1) == 1 ? 1 : x3d_env_int("X3D_DW_PD_S2", 2));
  if (dw_bwd_mx_launch(a, f->dtype, S, st) || dw_bwd_mxw_launch(a, f->dtype, S, st) || dw_bwd_mxg_launch(a, f->dtype, S, st)) {   // 14x14 stride-1 planes, bf16: both gradients on the matrix cores (dw_mx.hip)
    if (x3d_describe.out) return X3D_OK;
    X3D_LAUNCH_CHECK("dw3d_bwd");
    return X3D_OK;
  }
  if (S == 1 && cv > 0 && dw_bwd_s1_launch(a, f->dtype, SW, cv, (unsigned)grid, bd, st)) {   // strips of four, rows of whole vectors (56 x 56, 80 x 80, 40 x 40): dw_s1.hip
    if (x3d_describe.out) return X3D_OK;
    X3D_LAUNCH_CHECK("dw3d_bwd");
    return X3D_OK;
  }
  if (cv > 0 && dw_bwd_pk_launch(a, f->dtype, S, SW, st)) {   // 10..18-wide stride-1 planes: packed kernel (dw_pk.hip)
    if (x3d_describe.out) return X3D_OK;
    X3D_LAUNCH_CHECK("dw3d_bwd");
    return X3D_OK;
  }
  // stride 2, rows that only admit narrow vectors (28 x 28: 8 bytes, 14 x 14: 4 bytes): the deep-prefetch kernel with FLAT
  // 16-byte staging vectors when those bring the tile down to one vector per thread.  X3D_DW_PDFLAT=0: A/B hook.
  int pd_cv = cv, pd_nsv = nsv;
  if (S == 2 && pd > 1 && cv > 0 && nsv > 1 && x3d_env_int("X3D_DW_PDFLAT", 1) != 0) {
    const int va = 16 / (int)sizeof(T), vb = va / 2;
    if (a.g.W >= va && a.g.Wo >= vb && dw_nsv_flat(a.g.RIN, a.g.W, va, bd) <= 1 && dw_nsv_flat(a.RB, a.g.Wo, vb, bd) <= 1) {
      pd_cv = -va; pd_nsv = 1;
    }
  }
  // ... and with ragged own strips (left pad of an odd row, strips cut by the row end: 39 -> 20, 78 -> 39), 16-bit storage
  bool pd_ragged = false;
  if (S == 2 && sizeof(T) == 2 && pd > 1 && cv <= 0 && SW <= 2 && x3d_env_int("X3D_DW_PDFLAT", 1) != 0) {
    if (a.g.W >= 8 && a.g.Wo >= 4 && dw_nsv_flat(a.g.RIN, a.g.W, 8, bd) <= 1 && dw_nsv_flat(a.RB, a.g.Wo, 4, bd) <= 1) {
      pd_cv = -108; pd_nsv = 1; pd_ragged = true;
    }
  }
  if (pd > 1 && (cv > 0 || pd_ragged) && pd_nsv <= 1 && (long long)f->T * f->H * f->W * (long long)sizeof(T) < (1ll << 30)) {
    if (dw_bwd_pd_launch(a, f->dtype, S, SW, pd_cv, pd, (unsigned)grid, bd, lds, st)) {
      if (x3d_describe.out) return X3D_OK;
      X3D_LAUNCH_CHECK("dw3d_bwd");
      return X3D_OK;
    }
  }
  switch (SW) {
    case 4:
      if constexpr (S == 1) { dw_bwd_launch_nsv<T, S, 4>(a, nsv, cv, (unsigned)grid, bd, lds, st); break; }
    case 2: dw_bwd_launch_nsv<T, S, 2>(a, nsv, cv, (unsigned)grid, bd, lds, st); break;
    default: dw_bwd_launch_nsv<T, S, 1>(a, nsv, cv, (unsigned)grid, bd, lds, st); break;
  }
  if (x3d_describe.out) return X3D_OK;
  X3D_LAUNCH_CHECK("dw3d_bwd");
  return X3D_OK;
}

extern "C" int x3d_dw3d_bwd(const x3d_dw3d_bwd_args* f, void* stream) {
  X3D_REQUIRE(f && f->dv && f->braw && f->coef_nc && f->araw && f->a_scale_shift && f->w && f->ga &&
                  f->a_sums && f->dw, "dw3d_bwd: null pointer");
  X3D_REQUIRE(f->stride == 1 || f->stride == 2, "dw3d_bwd: stride must be 1 or 2");
  X3D_REQUIRE(f->N > 0 && f->C > 0 && f->T > 0 && f->H > 0 && f->W > 0, "dw3d_bwd: bad extents");
  X3D_REQUIRE(x3d_dtype_ok(f->dtype), "dw3d_bwd: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  if (f->dtype == X3D_F32)
    return f->stride == 1 ? dw_bwd_launch<float, 1>(f, st) : dw_bwd_launch<float, 2>(f, st);
  if (f->dtype == X3D_F16)
    return f->stride == 1 ? dw_bwd_launch<f16, 1>(f, st) : dw_bwd_launch<f16, 2>(f, st);
  return f->stride == 1 ? dw_bwd_launch<bf16, 1>(f, st) : dw_bwd_launch<bf16, 2>(f, st);
}
