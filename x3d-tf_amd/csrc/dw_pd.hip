// Channelwise 3x3x3 convolution, DEEP-PREFETCH variants of dw_fwd.hip / dw_bwd.hip for small planes.
//
// Why: one workgroup streams the T planes of one (n, c) channel.  At 14x14 (7x7) a plane is 392 (98) bytes per
// tensor, so with one plane in flight a CU holds ~9 KB of outstanding loads where Little's law wants ~45 KB
// (8 TB/s / 256 CUs x ~1.5 us): the one-plane-ahead kernels sit at 2.1 (1.6) TB/s on those layers, parked on
// s_waitcnt.  Here PD planes are in flight per workgroup.  Three things make that work on CDNA:
//   * the PD register slots are compile-time (T loop unrolled by PD), never a runtime-indexed register set;
//   * vmcnt retires IN ORDER and the compiler's wait for slot d is "at most k younger operations outstanding",
//     with k counted over operations that are issued unconditionally -- so every global access of the loop is an
//     unconditional bounds-checked buffer instruction (out-of-range = lanes without a staging vector, rows outside
//     the image, planes past T: zero / dropped, no memory traffic), and the prologue issues the same
//     load / store sequence as a steady-state iteration so the counts agree at the loop header;
//   * the thread's own araw strip (needed when its dA plane is emitted) travels with the prefetched planes instead
//     of being loaded one iteration ahead: a younger load ahead of its use would drag the wait past every
//     prefetched plane.
// Arithmetic, accumulation order and results are identical to the one-plane-ahead kernels.
#include "dw_common.h"

// ================================================================================================
// forward
// ================================================================================================
template <typename T, int S, int SW, int CV, int PD>
__global__ __launch_bounds__(256) void dw3d_fwd_pd_kernel(const DwFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  constexpr int WIN = (SW - 1) * S + 3;
  constexpr int EB = (int)sizeof(T);
  const int plane_sz = g.RIN * g.LP;
  float* scratch = lds + plane_sz;

  int b = blockIdx.x;
  const int tile = __builtin_amdgcn_readfirstlane(b % g.ntile_h); b /= g.ntile_h;
  const int c = __builtin_amdgcn_readfirstlane(b % g.C);
  const int n = __builtin_amdgcn_readfirstlane(b / g.C);
  const int h0 = tile * g.TH;
  const int th_here = min(g.TH, g.Ho - h0);
  const int r = threadIdx.x / g.nstrips, sidx = threadIdx.x - r * g.nstrips;
  const bool active = r < th_here;
  const int ho = h0 + r, wo0 = sidx * SW;

  for (int i = threadIdx.x; i < plane_sz; i += blockDim.x) lds[i] = 0.f;

  // weights as (kt=2, kt=1) pairs for the packed planes (out[t-1], out[t]) and kt=0 for the third (out[t+1])
  v2f w21[3][3];
  float w0[3][3];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    w21[k / 3][k % 3] = (v2f){a.w[c * 27 + 18 + k], a.w[c * 27 + 9 + k]};
    w0[k / 3][k % 3] = a.w[c * 27 + k];
  }
  float sc = 1.f, sh = 0.f;
  if (a.bn.stats) bn_fold_channel(a.bn, c, n == 0 && tile == 0 && threadIdx.x == 0, sc, sh, g.C);   // BN finalize folded in
  else if (a.ss) { sc = a.ss[c * 2]; sh = a.ss[c * 2 + 1]; }
  const int act = a.act;
  auto xf = [=](float v) {
    float u = sc * v + sh;
    return act == X3D_ACT_RELU ? fmaxf(u, 0.f) : u;
  };

  const int iplB = g.H * g.W * EB, oplB = g.Ho * g.Wo * EB;   // plane sizes in bytes
  const long long chan = (long long)n * g.C + c;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.x + chan * g.T * g.H * g.W, 0, g.T * iplB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc((T*)a.y + chan * g.T * g.Ho * g.Wo, 0, g.T * oplB, 0x00020000);
  const int row0 = h0 * S - g.ph;

  // staging map: ONE vector of CV elements per thread (host guarantees RIN * W / CV <= blockDim)
  int gX = DW_OOB, lX = 0;
  bool okX = false;
  {
    const int nvr = g.W / CV, v = threadIdx.x;
    if (v < g.RIN * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = row0 + lr;
      if (hi >= 0 && hi < g.H) { okX = true; gX = (hi * g.W + jv * CV) * EB; lX = lr * g.LP + g.pw + jv * CV; }
    }
  }
  const int oY = active ? (ho * g.Wo + wo0) * EB : DW_OOB;   // Wo % SW == 0 (host): whole strips only

  Raw raw[PD];
  float s1 = 0.f, s2 = 0.f;
  auto store_plane = [&](int t, bool live, const float (&v)[SW]) {   // !live: nothing stored, nothing summed
    Raw o;
    raw_pack<T, SW>(o, v);
    raw_bstore<SW * EB>(o, rsY, live ? oY + t * oplB : DW_OOB, 0);
#pragma unroll
    for (int i = 0; i < SW; i++) {   // inactive threads hold zeros
      const float u = live ? v[i] : 0.f;
      s1 += u;
      s2 += u * u;
    }
  };

  v2f acc01[SW], acc2p[(SW + 1) / 2];   // (out[t-1], out[t]) per output, out[t+1] as pairs over outputs
  float fin[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) { acc01[i] = (v2f){0.f, 0.f}; fin[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < (SW + 1) / 2; j++) acc2p[j] = (v2f){0.f, 0.f};

  // prologue: planes 0..PD-1 in flight, each followed by a (dropped) store like a steady-state iteration
#pragma unroll
  for (int d = 0; d < PD; d++) {
    raw_bload<CV * EB>(raw[d], rsX, gX + d * iplB, 0);
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
    raw_bstore<SW * EB>(z, rsY, DW_OOB, 0);
  }

  for (int t0 = 0; t0 < g.T; t0 += PD) {
#pragma unroll
    for (int d = 0; d < PD; d++) {
      const int t = t0 + d;
      if (t >= g.T) break;
      __syncthreads();  // zero-fill / previous plane's readers done
      if (okX) {
        float* dst = lds + lX;
#pragma unroll
        for (int e = 0; e < CV; e++) dst[e] = xf(raw_get<T>(raw[d], e));
      }
      __syncthreads();
      raw_bload<CV * EB>(raw[d], rsX, gX + (t + PD) * iplB, 0);   // past T: out of range, nothing moves
      store_plane(t - 2, t >= 2, fin);                            // plane t-2 (finished last iteration)
      if (active) {
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
          float win[WIN];
          const float* row = lds + (r * S + kh) * g.LP + wo0 * S;
          lds_window<WIN, (SW * S >= 4 ? 4 : SW * S)>(row, win);
          dw_taps_row<S, SW, WIN>(win, w21[kh], w0[kh], acc01, acc2p);
        }
      }
      dw_rotate<SW>(fin, acc01, acc2p);
    }
  }
  store_plane(g.T - 2, g.T >= 2, fin);
  float last[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) last[i] = acc01[i].x;
  store_plane(g.T - 1, true, last);

  if (a.stats || a.pool) {
    float red[2] = {s1, s2};
    block_sum<2>(red, scratch);
    if (threadIdx.x == 0) {
      if (a.stats) {
        double* sp = stats_replica(a.stats, g.C, (unsigned)(n * g.ntile_h + tile));
        atomic_add_d(&sp[c * 2], (double)red[0]);
        atomic_add_d(&sp[c * 2 + 1], (double)red[1]);
      }
      if (a.pool) atomic_add_d(&a.pool[(long long)n * g.C + c], (double)red[0]);
    }
  }
}

// ================================================================================================
// backward (data + weight fused); formulas in dw_bwd.hip
// ================================================================================================
// waves per SIMD the register allocator must keep (amdgpu-waves-per-eu): the narrow instantiations fit 128 VGPRs
// without spilling and are only worth running at 4 waves; the wide ones would spill
#ifndef DW_S1_OCC
#define DW_S1_OCC 4
#endif
template <int S, int SW, int CV> struct BwdPdWaves {
  static constexpr int v = CV < 0 ? 1 : ((S == 1 && SW == 2 && CV <= 2) ? DW_S1_OCC : ((SW == 1 && CV <= 4) ? 4 : 1));
};
// RO = 1 (16-bit types, flat staging): RAGGED OWN STRIPS -- the NA dA columns a thread owns may start at column -1 (TF-SAME
// left pad of an odd row, 39 -> 20) or run past the row end (78 -> 39 with strips of 2 outputs).  The araw strip is then
// loaded from the nearest position inside the row and moved into place with one 32/64-bit shift pair; whole strips are
// stored as one (unaligned) vector, cut strips element by element -- every instruction unconditional, the offsets of the
// accesses that must not happen point past the buffer.
// (flat staging, strips of 2, 16-bit storage: 162-169 VGPRs -- held to the 168 of three waves per SIMD)
template <typename T, int S, int SW, int CV, int PD, int RO = 0>
__global__ __launch_bounds__(256, ((CV < 0 && SW == 2 && sizeof(T) == 2) ? 3 : BwdPdWaves<S, SW, CV>::v)) void dw3d_bwd_pd_kernel(const DwBwdArgs a) {
  static_assert(RO == 0 || (CV < 0 && sizeof(T) == 2), "ragged own strips: flat staging, 16-bit storage");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  constexpr int WIN = (SW - 1) * S + 3;          // act window columns
  constexpr int BW = (S == 1) ? SW + 2 : SW + 1; // dB window columns
  constexpr int BR = (S == 1) ? 3 : 2;           // dB window rows
  constexpr int NA = (S == 1) ? SW : 2 * SW;     // dA columns owned per row
  constexpr int NR = (S == 1) ? 1 : 2;           // dA rows owned
  constexpr bool DEFER = (S == 1);               // emit a finished dA plane one iteration late (register budget, see dw_bwd.hip)
  // CV < 0: FLAT staging (FlatMap, dw_common.h) in vectors of -CV elements cut from the tile's contiguous run of rows --
  // whole 16-byte loads on planes whose rows are not (28 x 28: 56-byte rows, 14 x 14: 28-byte rows), so that one vector
  // per thread covers the tile
  constexpr bool FLAT = CV < 0;
  constexpr int VA = FLAT ? -CV : CV, VB = (S == 1) ? VA : (VA > 1 ? VA / 2 : 1);
  constexpr int EB = (int)sizeof(T);
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

  // staging maps: ONE vector per thread and tensor (host guarantees it)
  int gA = DW_OOB, lA = 0, gB = DW_OOB, lB = 0, wrA = VA, wrB = VB;
  bool okA = false, okB = false;
  if constexpr (FLAT) {
    {
      const int rlo = rowA0 > 0 ? rowA0 : 0, rhi = (rowA0 + g.RIN < g.H) ? rowA0 + g.RIN : g.H;
      const int span = (rhi - rlo) * g.W, nvec = (span + VA - 1) / VA, v = threadIdx.x;
      if (v < nvec) {
        int e = v * VA;
        if (e > span - VA) e = span - VA;
        const int lr = e / g.W, col = e - lr * g.W;
        okA = true; gA = (rlo * g.W + e) * EB; lA = (rlo - rowA0 + lr) * g.LP + g.pw + col; wrA = g.W - col;
      }
    }
    {
      const int rb0 = h0 - 1;
      const int rlo = rb0 > 0 ? rb0 : 0, rhi = (rb0 + a.RB < g.Ho) ? rb0 + a.RB : g.Ho;
      const int span = (rhi - rlo) * g.Wo, nvec = (span + VB - 1) / VB, v = threadIdx.x;
      if (v < nvec) {
        int e = v * VB;
        if (e > span - VB) e = span - VB;
        const int lr = e / g.Wo, col = e - lr * g.Wo;
        okB = true; gB = (rlo * g.Wo + e) * EB; lB = (rlo - rb0 + lr) * a.LPB + 1 + col; wrB = g.Wo - col;
      }
    }
  } else {
  {
    const int nvr = g.W / VA, v = threadIdx.x;
    if (v < g.RIN * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = rowA0 + lr;
      if (hi >= 0 && hi < g.H) { okA = true; gA = (hi * g.W + jv * VA) * EB; lA = lr * g.LP + g.pw + jv * VA; }
    }
  }
  {  // dB plane: lds row 0 <-> output row h0-1, col 0 <-> col -1
    const int nvr = g.Wo / VB, v = threadIdx.x;
    if (v < a.RB * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = h0 - 1 + lr;
      if (hi >= 0 && hi < g.Ho) { okB = true; gB = (hi * g.Wo + jv * VB) * EB; lB = lr * a.LPB + 1 + jv * VB; }
    }
  }
  }
  // image rows / cols owned for dA (NA contiguous, NA-aligned columns per row: host checks W % NA == 0, pw == 0 for S == 2)
  const int hA = (S == 1) ? ho : ho * 2 - g.ph;
  const int wA0 = (S == 1) ? wo0 : wo0 * 2 - g.pw;
  int oOwn[NR];
  bool okOwn[NR];
  // ragged own strips: ws = first column actually loaded (strip moved inside the row), shr / shl = bit shifts that put
  // column wA0 + i back at element i, cmask = bit i set when column wA0 + i exists, full = the whole strip exists
  const int ws = RO ? (wA0 < 0 ? 0 : (wA0 + NA > g.W ? g.W - NA : wA0)) : wA0;
  const int shr = RO ? (wA0 > ws ? 16 * (wA0 - ws) : 0) : 0, shl = RO ? (ws > wA0 ? 16 * (ws - wA0) : 0) : 0;
  unsigned cmask = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) if (!RO || (wA0 + i >= 0 && wA0 + i < g.W)) cmask |= 1u << i;
  const bool full = cmask == (1u << NA) - 1;
  int oSt[NR];   // store offset of column wA0 (element stores add 2 * i)
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int h = hA + q;
    okOwn[q] = active && h >= 0 && h < g.H;
    oOwn[q] = okOwn[q] ? (h * g.W + ws) * EB : DW_OOB;
    oSt[q] = okOwn[q] ? (h * g.W + wA0) * EB : DW_OOB;
  }
  auto place = [&](Raw& o) {   // loaded strip -> element i = column wA0 + i (columns that do not exist: zero bits)
    if constexpr (RO) {
      if constexpr (NA * EB == 8) {
        unsigned long long v = ((unsigned long long)o.w[1] << 32) | o.w[0];
        v = (v >> shr) << shl;
        o.w[0] = (uint32_t)v; o.w[1] = (uint32_t)(v >> 32);
      } else {
        o.w[0] = (o.w[0] >> shr) << shl;
      }
    }
  };

  struct Slot { Raw A, D, R, O[NR]; };
  Slot slot[PD];
  auto issue = [&](int t, Slot& s) {
    raw_bload<VA * EB>(s.A, rsA, gA + t * iplB, 0);
    raw_bload<VB * EB>(s.D, rsD, gB + t * oplB, 0);
    raw_bload<VB * EB>(s.R, rsR, gB + t * oplB, 0);
#pragma unroll
    for (int q = 0; q < NR; q++) raw_bload<NA * EB>(s.O[q], rsA, oOwn[q] + t * iplB, 0);
  };
#ifdef X3D_EXPERIMENTS
  const bool x_nomath = a.exp & 1, x_noload = a.exp & 2, x_noemit = a.exp & 4;
#else
  constexpr bool x_nomath = false, x_noload = false, x_noemit = false;
#endif
  if (x_noload) {
    gA = DW_OOB; gB = DW_OOB;
#pragma unroll
    for (int q = 0; q < NR; q++) oOwn[q] = DW_OOB;
  }

  float dA0[NR][NA], dA1[NR][NA], dA2[NR][NA], fin[NR][NA];
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
  Raw own1[NR], own2[NR];   // araw strips of planes t-1 / t-2 (the slot itself is refilled right after staging)
#pragma unroll
  for (int q = 0; q < NR; q++) { own1[q].w[0] = own1[q].w[1] = own1[q].w[2] = own1[q].w[3] = 0u; own2[q] = own1[q]; }

  auto emit = [&](int t, bool live, const float (&v)[NR][NA], const Raw (&own)[NR]) {   // !live: nothing stored / summed
    if (x_noload) live = false;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      float gv[NA];
      if (x_noemit) {
#pragma unroll
        for (int i = 0; i < NA; i++) gv[i] = v[q][i];
      } else {
#pragma unroll
      for (int i = 0; i < NA; i++) {
        const float av = raw_get<T>(own[q], i);
        gv[i] = (live && okOwn[q] && (!RO || ((cmask >> i) & 1)) && sc * av + sh > 0.f) ? v[q][i] : 0.f;
        s1 += gv[i];
        s2 += gv[i] * av;
      }
      }
      Raw o;
      raw_pack<T, NA>(o, gv);
      if constexpr (RO) {
        const int base = oSt[q] + t * iplB;
        raw_bstore<NA * EB>(o, rsG, (live && full) ? base : DW_OOB, 0);
#pragma unroll
        for (int i = 0; i < NA; i++) {
          Raw e1; e1.w[0] = (o.w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
          raw_bstore<EB>(e1, rsG, (live && !full && okOwn[q] && ((cmask >> i) & 1)) ? base + EB * i : DW_OOB, 0);
        }
      } else {
        raw_bstore<NA * EB>(o, rsG, live ? oOwn[q] + t * iplB : DW_OOB, 0);
      }
    }
  };

  // prologue: the same load / store sequence as a steady-state iteration (stores dropped)
#pragma unroll
  for (int d = 0; d < PD; d++) {
    issue(d, slot[d]);
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      raw_bstore<NA * EB>(z, rsG, DW_OOB, 0);
      if constexpr (RO) {
#pragma unroll
        for (int i = 0; i < NA; i++) raw_bstore<EB>(z, rsG, DW_OOB, 0);
      }
    }
  }

  for (int t0 = 0; t0 < g.T; t0 += PD) {
#pragma unroll
    for (int d = 0; d < PD; d++) {
      const int t = t0 + d;
      if (t >= g.T) break;
      __syncthreads();
      if (okA) {
        float* dst = Al + lA;
        if constexpr (FLAT) flat_commit<T, VA>(dst, wrA, g.LP - g.W, slot[d].A, af);
        else {
#pragma unroll
          for (int e = 0; e < VA; e++) dst[e] = af(raw_get<T>(slot[d].A, e));
        }
      }
      if (okB) {
        float* dst = Bl + lB;
        if constexpr (FLAT) flat_commit2<T, VB>(dst, wrB, a.LPB - g.Wo, slot[d].D, slot[d].R, bf);
        else {
#pragma unroll
          for (int e = 0; e < VB; e++) dst[e] = bf(raw_get<T>(slot[d].D, e), raw_get<T>(slot[d].R, e));
        }
      }
      Raw own0[NR];
#pragma unroll
      for (int q = 0; q < NR; q++) { own0[q] = slot[d].O[q]; place(own0[q]); }
      __syncthreads();
      issue(t + PD, slot[d]);
      if constexpr (DEFER) emit(t - 2, t >= 2, fin, own2);   // plane t-2, completed at the end of iteration t-1
      if (active && !x_nomath) {
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
        float dBo[SW];
#pragma unroll
        for (int i = 0; i < SW; i++) dBo[i] = winB[1][i + 1];

        // scalar FMAs: the packed form (dw_common.h) costs ~20 VGPRs here and a wave of occupancy -- measured slower
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++) {
#pragma unroll
            for (int i = 0; i < SW; i++) {
              dW[9 + kh * 3 + kw] += dBo[i] * winA[kh][i * S + kw];
              dW[18 + kh * 3 + kw] += dB_prev[i] * winA[kh][i * S + kw];
              dW[kh * 3 + kw] += dBo[i] * winA_prev[kh][i * S + kw];
            }
          }

        if constexpr (S == 1) {
#pragma unroll
          for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int kw = 0; kw < 3; kw++)
#pragma unroll
              for (int i = 0; i < SW; i++) {
                const float v = winB[2 - kh][i + 2 - kw];
                dA0[0][i] += wgt[kh * 3 + kw] * v;
                dA1[0][i] += wgt[9 + kh * 3 + kw] * v;
                dA2[0][i] += wgt[18 + kh * 3 + kw] * v;
              }
        } else {
#pragma unroll
          for (int i = 0; i < SW; i++) {
            const float b11 = winB[1][i + 1], b10 = winB[1][i], b01 = winB[0][i + 1], b00 = winB[0][i];
#pragma unroll
            for (int kt = 0; kt < 3; kt++) {
              const float* wk = &wgt[kt * 9];
              const float eA = wk[0] * b11 + wk[2] * b10 + wk[6] * b01 + wk[8] * b00;
              const float eB = wk[1] * b11 + wk[7] * b01;
              const float eC = wk[3] * b11 + wk[5] * b10;
              const float eD = wk[4] * b11;
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
      if constexpr (!DEFER) emit(t - 1, t >= 1, dA0, own1);   // plane t-1 is complete now
#pragma unroll
      for (int q = 0; q < NR; q++) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
          if constexpr (DEFER) fin[q][i] = dA0[q][i];
          dA0[q][i] = dA1[q][i]; dA1[q][i] = dA2[q][i]; dA2[q][i] = 0.f;
        }
        own2[q] = own1[q];
        own1[q] = own0[q];
      }
    }
  }
  // after the loop: own1 = strip of plane T-1, own2 = plane T-2; fin = dA plane T-2 (DEFER), dA0 = plane T-1
  if constexpr (DEFER) emit(g.T - 2, g.T >= 2, fin, own2);
  emit(g.T - 1, true, dA0, own1);

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

// ---- stride-1 variant with register ROLES (see the comment at the accumulators) ----
template <typename T, int SW, int CV, int PD, int UN>
__global__ __launch_bounds__(256, (BwdPdWaves<1, SW, CV>::v)) void dw3d_bwd_pd_s1_kernel(const DwBwdArgs a) {
  constexpr int S = 1;
  static_assert(UN % 6 == 0 && UN % PD == 0, "roles have periods 2 (windows) and 3 (planes); slots period PD");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  constexpr int WIN = (SW - 1) * S + 3;          // act window columns
  constexpr int BW = (S == 1) ? SW + 2 : SW + 1; // dB window columns
  constexpr int BR = (S == 1) ? 3 : 2;           // dB window rows
  constexpr int NA = (S == 1) ? SW : 2 * SW;     // dA columns owned per row
  constexpr int NR = (S == 1) ? 1 : 2;           // dA rows owned
  constexpr bool DEFER = (S == 1);               // emit a finished dA plane one iteration late (register budget, see dw_bwd.hip)
  constexpr int VA = CV, VB = (S == 1) ? CV : (CV > 1 ? CV / 2 : 1);
  constexpr int EB = (int)sizeof(T);
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

  // staging maps: ONE vector per thread and tensor (host guarantees it)
  int gA = DW_OOB, lA = 0, gB = DW_OOB, lB = 0;
  bool okA = false, okB = false;
  {
    const int nvr = g.W / VA, v = threadIdx.x;
    if (v < g.RIN * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = rowA0 + lr;
      if (hi >= 0 && hi < g.H) { okA = true; gA = (hi * g.W + jv * VA) * EB; lA = lr * g.LP + g.pw + jv * VA; }
    }
  }
  {  // dB plane: lds row 0 <-> output row h0-1, col 0 <-> col -1
    const int nvr = g.Wo / VB, v = threadIdx.x;
    if (v < a.RB * nvr) {
      const int lr = v / nvr, jv = v - lr * nvr, hi = h0 - 1 + lr;
      if (hi >= 0 && hi < g.Ho) { okB = true; gB = (hi * g.Wo + jv * VB) * EB; lB = lr * a.LPB + 1 + jv * VB; }
    }
  }
  // image rows / cols owned for dA (NA contiguous, NA-aligned columns per row: host checks W % NA == 0, pw == 0 for S == 2)
  const int hA = (S == 1) ? ho : ho * 2 - g.ph;
  const int wA0 = (S == 1) ? wo0 : wo0 * 2 - g.pw;
  int oOwn[NR];
  bool okOwn[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int h = hA + q;
    okOwn[q] = active && h >= 0 && h < g.H;
    oOwn[q] = okOwn[q] ? (h * g.W + wA0) * EB : DW_OOB;
  }

  struct Slot { Raw A, D, R, O[NR]; };
  Slot slot[PD];
  auto issue = [&](int t, Slot& s) {
    raw_bload<VA * EB>(s.A, rsA, gA + t * iplB, 0);
    raw_bload<VB * EB>(s.D, rsD, gB + t * oplB, 0);
    raw_bload<VB * EB>(s.R, rsR, gB + t * oplB, 0);
#pragma unroll
    for (int q = 0; q < NR; q++) raw_bload<NA * EB>(s.O[q], rsA, oOwn[q] + t * iplB, 0);
  };

  // Registers with ROLES instead of copies: the T loop is unrolled by UN (a multiple of 6), so which register set is
  // "the previous plane's window" (period 2) and which accumulator is "plane t-1 / t / t+1" (period 3) are compile-time
  // facts of each unrolled iteration -- the ~20 v_mov per plane of the copying form (window -> previous window,
  // dA1 -> dA0, dA2 -> dA1, dA0 -> fin) disappear.  Slots: plane t sits in slot t % PD.
  float dAr[3][SW];            // dAr[p % 3] = gradient plane p while it is being accumulated / waiting for its emit
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int i = 0; i < SW; i++) dAr[k][i] = 0.f;
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
  float s1 = 0.f, s2 = 0.f;
  Raw own1, own2;   // araw strips of planes t-1 / t-2 (the slot itself is refilled right after staging)
  own1.w[0] = own1.w[1] = own1.w[2] = own1.w[3] = 0u;
  own2 = own1;

  auto emit = [&](int t, bool live, const float (&v)[SW], const Raw& own) {   // !live: nothing stored / summed
    float gv[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) {
      const float av = raw_get<T>(own, i);
      gv[i] = (live && okOwn[0] && sc * av + sh > 0.f) ? v[i] : 0.f;
      s1 += gv[i];
      s2 += gv[i] * av;
    }
    Raw o;
    raw_pack<T, SW>(o, gv);
    raw_bstore<SW * EB>(o, rsG, live ? oOwn[0] + t * iplB : DW_OOB, 0);
  };

  // prologue: the same load / store sequence as a steady-state iteration (stores dropped)
#pragma unroll
  for (int d = 0; d < PD; d++) {
    issue(d, slot[d]);
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
    raw_bstore<SW * EB>(z, rsG, DW_OOB, 0);
  }

  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      if (t >= g.T) break;
      const int sl = d % PD, cur = d & 1, prv = cur ^ 1;           // compile-time after unrolling (t0 % UN == 0)
      const int pm1 = (d + 2) % 3, p0 = d % 3, pp1 = (d + 1) % 3;  // accumulators of planes t-1, t, t+1
      __syncthreads();
      if (okA) {
        float* dst = Al + lA;
#pragma unroll
        for (int e = 0; e < VA; e++) dst[e] = af(raw_get<T>(slot[sl].A, e));
      }
      if (okB) {
        float* dst = Bl + lB;
#pragma unroll
        for (int e = 0; e < VB; e++) dst[e] = bf(raw_get<T>(slot[sl].D, e), raw_get<T>(slot[sl].R, e));
      }
      const Raw own0 = slot[sl].O[0];
      __syncthreads();
      issue(t + PD, slot[sl]);
      // plane t-2 was completed at the end of iteration t-1; it lives in the accumulator that plane t+1 takes over now
      emit(t - 2, t >= 2, dAr[pp1], own2);
#pragma unroll
      for (int i = 0; i < SW; i++) dAr[pp1][i] = 0.f;
      if (active) {
        float winB[BR][BW];
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
          const float* row = Al + (r + kh) * g.LP + wo0;
          lds_window<WIN, (SW >= 4 ? 4 : SW)>(row, winA2[cur][kh]);
        }
#pragma unroll
        for (int q = 0; q < BR; q++) {
          const float* row = Bl + (r + q) * a.LPB + wo0;
          lds_window<BW, (SW >= 4 ? 4 : SW)>(row, winB[q]);
        }
#pragma unroll
        for (int i = 0; i < SW; i++) dBs[cur][i] = winB[1][i + 1];

#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++) {
#pragma unroll
            for (int i = 0; i < SW; i++) {
              dW[9 + kh * 3 + kw] += dBs[cur][i] * winA2[cur][kh][i + kw];
              dW[18 + kh * 3 + kw] += dBs[prv][i] * winA2[cur][kh][i + kw];
              dW[kh * 3 + kw] += dBs[cur][i] * winA2[prv][kh][i + kw];
            }
          }
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int i = 0; i < SW; i++) {
              const float v = winB[2 - kh][i + 2 - kw];
              dAr[pm1][i] += wgt[kh * 3 + kw] * v;
              dAr[p0][i] += wgt[9 + kh * 3 + kw] * v;
              dAr[pp1][i] += wgt[18 + kh * 3 + kw] * v;
            }
      }
      own2 = own1;
      own1 = own0;
    }
  }
  // after the loop (T planes staged): plane T-2 waits in dAr[(T-2) % 3] with its strip in own2, plane T-1 in dAr[(T-1) % 3] / own1
  {
    const int m2 = (g.T + 1) % 3, m1 = (g.T + 2) % 3;   // (T-2) % 3, (T-1) % 3 for T >= 1
    float v2[SW], v1[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) {
      v2[i] = m2 == 0 ? dAr[0][i] : (m2 == 1 ? dAr[1][i] : dAr[2][i]);
      v1[i] = m1 == 0 ? dAr[0][i] : (m1 == 1 ? dAr[1][i] : dAr[2][i]);
    }
    emit(g.T - 2, g.T >= 2, v2, own2);
    emit(g.T - 1, true, v1, own1);
  }

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

// ================================================================================================
// dispatch.  Depth 4 for strips of 1 / 2 outputs (rows of < 20 outputs), depth 2 for strips of 4 (stride 1).
// ================================================================================================
int dw_pick_pd(int SW) {   // depth 4 for strips of 1 / 2 outputs; wider strips keep the one-plane-ahead kernels
  if (x3d_env_int("X3D_DW_PD", 4) <= 1) return 1;
  return SW <= 2 ? 4 : 1;
}

#define DW_PD_DESCRIBE(KIND)                                                                                   \
  if (x3d_describe.out) {                                                                                      \
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_" KIND "_pd_kernel<%s, %d, %d, %d, %d>",                \
             TypeName<T>::v, S, SW, CV, PD);                                                \
    return true;                                                                                               \
  }

template <typename T, int S, int SW, int CV, int PD>
static bool fwd_go(const DwFwdArgs& a, unsigned grid, int bd, size_t lds, hipStream_t st) {
  DW_PD_DESCRIBE("fwd")
  hipLaunchKernelGGL((dw3d_fwd_pd_kernel<T, S, SW, CV, PD>), dim3(grid), dim3(bd), lds, st, a);
  return true;
}
template <typename T, int S, int SW, int CV, int PD>
static bool bwd_go(const DwBwdArgs& a, unsigned grid, int bd, size_t lds, hipStream_t st) {
  // stride 1, strips of 2 (14x14 / 10x10 planes): the register-role form at depth 3 (164 -> 155 us at 14x14).  Strips of 1
  // (7x7) keep the copying form at depth 4: there the role form needs the 128-VGPR cap's spills and was 0.7x.
  // X3D_DW_ROLES=0: copying form everywhere (A/B hook).
  const int roles = x3d_env_int("X3D_DW_ROLES", 1);
  if constexpr (S == 1 && SW == 2) {
    if (roles != 0) {
      if (x3d_describe.out) {
        snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_pd_s1_kernel<%s, %d, %d, %d, %d>", TypeName<T>::v,
                 SW, CV, 3, 6);
        return true;
      }
      hipLaunchKernelGGL((dw3d_bwd_pd_s1_kernel<T, SW, CV, 3, 6>), dim3(grid), dim3(bd), lds, st, a);
      return true;
    }
  }
  if (x3d_describe.out) {   // the name rocprofv3 prints: all six template arguments (RO = 0)
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_pd_kernel<%s, %d, %d, %d, %d, 0>", TypeName<T>::v, S, SW, CV, PD);
    return true;
  }
  hipLaunchKernelGGL((dw3d_bwd_pd_kernel<T, S, SW, CV, PD>), dim3(grid), dim3(bd), lds, st, a);
  return true;
}

// (S, SW) -> depth: {1,2} -> 4, 4 -> 2.  CV in {1,2,4,8 (bf16 only)}.
#define DW_PD_CV(GO, T, S, SW, PD)                                                       \
  switch (cv) {                                                                          \
    case 8: if constexpr (sizeof(T) == 2) return GO<T, S, SW, 8, PD>(a, grid, bd, lds, st); else return false; \
    case 4: return GO<T, S, SW, 4, PD>(a, grid, bd, lds, st);                            \
    case 2: return GO<T, S, SW, 2, PD>(a, grid, bd, lds, st);                            \
    case 1: return GO<T, S, SW, 1, PD>(a, grid, bd, lds, st);                            \
    default: return false;                                                               \
  }

template <typename T>
static bool fwd_pd_t(const DwFwdArgs& a, int S, int SW, int cv, int pd, unsigned grid, int bd, size_t lds, hipStream_t st) {
  if (S == 1) {
    if (SW == 1 && pd == 4) { DW_PD_CV(fwd_go, T, 1, 1, 4) }
    if (SW == 2 && pd == 4) { DW_PD_CV(fwd_go, T, 1, 2, 4) }
    // SW == 4 at depth 2: measured slower than the one-plane-ahead kernel (255 vs 279 us at 56x56)
  } else {
    if (SW == 1 && pd == 4) { DW_PD_CV(fwd_go, T, 2, 1, 4) }
    if (SW == 2 && pd == 4) { DW_PD_CV(fwd_go, T, 2, 2, 4) }
  }
  return false;
}
template <typename T>
static bool bwd_pd_t(const DwBwdArgs& a, int S, int SW, int cv, int pd, unsigned grid, int bd, size_t lds, hipStream_t st) {
  if (S == 1) {
    if (SW == 1 && pd == 4) { DW_PD_CV(bwd_go, T, 1, 1, 4) }
    if (SW == 2 && pd == 4) { DW_PD_CV(bwd_go, T, 1, 2, 4) }
    // SW == 4 at depth 2: 220-250 VGPRs (2 waves) -- the one-plane-ahead kernel keeps those planes
  } else {
    // stride 2: 2x2 input quads per output put the depth-4 kernel at 160-190 VGPRs (2 waves): measured 0.84x;
    // depth 2 (136-144 VGPRs, 3 waves) is 1.11x the one-plane-ahead kernel on the 112x112 / 56x56 layers
    // flat staging + ragged own strips (cv = -100 - vector width; 16-bit storage)
    if constexpr (sizeof(T) == 2) {
      if (cv == -108 && pd == 2) {
        if (x3d_describe.out) {
          snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_pd_kernel<%s, 2, %d, -8, 2, 1>", TypeName<T>::v, SW);
          return true;
        }
        if (SW == 1) { hipLaunchKernelGGL((dw3d_bwd_pd_kernel<T, 2, 1, -8, 2, 1>), dim3(grid), dim3(bd), lds, st, a); return true; }
        if (SW == 2) { hipLaunchKernelGGL((dw3d_bwd_pd_kernel<T, 2, 2, -8, 2, 1>), dim3(grid), dim3(bd), lds, st, a); return true; }
      }
    }
    // flat staging (cv < 0: 16-byte vectors whatever the row length)
    if (cv == -(int)(16 / sizeof(T)) && pd == 2) {
      if (SW == 1) return bwd_go<T, 2, 1, -(int)(16 / sizeof(T)), 2>(a, grid, bd, lds, st);
      if (SW == 2) return bwd_go<T, 2, 2, -(int)(16 / sizeof(T)), 2>(a, grid, bd, lds, st);
    }
    if (SW == 1 && pd == 2) { DW_PD_CV(bwd_go, T, 2, 1, 2) }
    if (SW == 2 && pd == 2) { DW_PD_CV(bwd_go, T, 2, 2, 2) }
  }
  return false;
}

bool dw_fwd_pd_launch(const DwFwdArgs& a, int dtype, int S, int SW, int cv, int pd, unsigned grid, int bd,
                      size_t lds, hipStream_t st) {
  return dtype == X3D_BF16 ? fwd_pd_t<bf16>(a, S, SW, cv, pd, grid, bd, lds, st)
         : dtype == X3D_F16 ? fwd_pd_t<f16>(a, S, SW, cv, pd, grid, bd, lds, st)
                            : fwd_pd_t<float>(a, S, SW, cv, pd, grid, bd, lds, st);
}
bool dw_bwd_pd_launch(const DwBwdArgs& a, int dtype, int S, int SW, int cv, int pd, unsigned grid, int bd,
                      size_t lds, hipStream_t st) {
  if (S == 2 && dw_bwd_s2_launch(a, dtype, SW, cv, pd, grid, bd, lds, st)) return true;
  return dtype == X3D_BF16 ? bwd_pd_t<bf16>(a, S, SW, cv, pd, grid, bd, lds, st)
         : dtype == X3D_F16 ? bwd_pd_t<f16>(a, S, SW, cv, pd, grid, bd, lds, st)
                            : bwd_pd_t<float>(a, S, SW, cv, pd, grid, bd, lds, st);
}
