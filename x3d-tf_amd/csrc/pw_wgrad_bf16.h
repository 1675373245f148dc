// Pointwise-conv weight gradient for bf16 storage on v_mfma_f32_32x32x16_bf16.
//   dW[co][ci] += sum_p dYraw[co][p] * f(X)[ci][p]       M = Cout, N = Cin, K = points
// Both operands are [row][points] with the points contiguous -- exactly the K-contiguous fragment the
// MFMA wants (lane = row, 8 consecutive k), so both tiles are staged as they lie in HBM and read with
// ds_read_b128; no transpose anywhere.  Row pitch = 64 points + 8 (144 B) keeps the 16 lanes of a b128 group
// on distinct 16-byte slots.  When a layer has fewer 32x32 tiles than waves (stage-2 widths: 54x24 = 2
// tiles) the idle waves take a share of every 64-point step instead (split-K inside the workgroup); all
// partial tiles meet in the fp32 atomics on dW.
#pragma once
#include "common.h"

template <typename H, int VEC, int TPW, bool XPRO, bool STRIDED>
__global__ __launch_bounds__(256) void pw_wgrad_bf16_kernel(const PwWgradArgs a) {
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4; typedef typename HV<H>::x2 hx2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BP = 64, LP = BP + 8;
  const int rowsA = a.mt_per_group * 32, rowsB = a.nt_total * 32;
  H* As = (H*)smem_raw;        // [rowsA][LP]  dYraw
  H* Bs = As + rowsA * LP;        // [rowsB][LP]  f(X)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.y * rowsA;
  const int steps_per_n = (int)((a.P + BP - 1) / BP);
  const int chunks_per_n = (steps_per_n + a.steps_per_block - 1) / a.steps_per_block;
  const int n = blockIdx.x / chunks_per_n;
  const int chunk = blockIdx.x - n * chunks_per_n;
  const int s_begin = chunk * a.steps_per_block;
  const int s_end = min(s_begin + a.steps_per_block, steps_per_n);
  const int mt_here = min(a.mt_per_group, (a.Cout - co0 + 31) / 32);
  const int ntiles = mt_here * a.nt_total;
  // split-K over waves when there are fewer tiles than waves (only possible with TPW == 1)
  const int nks = (TPW == 1 && ntiles <= 2) ? 4 / ntiles : 1;   // 1, 2 or 4 k-parts per 64-point step
  const int ksteps = (BP / 16) / nks;                            // 16-wide MFMA k-steps per wave per step

  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

  constexpr int VPR = BP / VEC;
  for (int step = s_begin; step < s_end; ++step) {
    const long long p0 = (long long)step * BP;
    __syncthreads();
    for (int v = tid; v < rowsA * VPR; v += 256) {
      const int row = v / VPR, pv = v - row * VPR;
      const int co = co0 + row;
      const long long p = p0 + (long long)pv * VEC;
      float val[VEC];
      if (co < a.Cout && p < a.P) {
        const long long o = ((long long)n * a.Cout + co) * a.P + p;
        VecIO<T, VEC>::load((const T*)a.g + o, val);
        if (a.coef) {
          float y2[VEC];
          VecIO<T, VEC>::load((const T*)a.yraw + o, y2);
          const float A = a.coef[co * 4], B = a.coef[co * 4 + 1], C = a.coef[co * 4 + 2];
#pragma unroll
          for (int e = 0; e < VEC; e++) val[e] = A * val[e] + B * y2[e] + C;
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) val[e] = 0.f;
      }
      VecIO<H, VEC>::store(&As[row * LP + pv * VEC], val);
    }
    for (int v = tid; v < rowsB * VPR; v += 256) {
      const int row = v / VPR, pv = v - row * VPR;
      const long long p = p0 + (long long)pv * VEC;
      H* dst = &Bs[row * LP + pv * VEC];
      if (row < a.Cin && p < a.P) {
        if constexpr (!XPRO && !STRIDED && VEC == 8) {
          *(hx8*)dst = *(const hx8*)((const T*)a.x + ((long long)n * a.Cin + row) * a.Pin + p);
        } else {
          float val[VEC];
          if constexpr (STRIDED) {
            const long long hw = (long long)a.Ho * a.Wo;
            const long long t = p / hw;
            const int rem = (int)(p - t * hw);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const long long src = (t * a.H + (long long)ho * a.stride) * a.W + (long long)wo * a.stride;
            val[0] = to_f<T>(((const T*)a.x)[((long long)n * a.Cin + row) * a.Pin + src]);
          } else {
            VecIO<T, VEC>::load((const T*)a.x + ((long long)n * a.Cin + row) * a.Pin + p, val);
          }
          if constexpr (XPRO) {
            const float s = a.xcoef[row * 2], t = a.xcoef[row * 2 + 1];
            const float g = a.xgate ? a.xgate[(long long)n * a.Cin + row] : 1.0f;
#pragma unroll
            for (int e = 0; e < VEC; e++) val[e] = (s * val[e] + t) * g;
            act_vec<VEC>(val, a.xact);
          }
          VecIO<H, VEC>::store(dst, val);
        }
      } else {
        float z[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) z[e] = 0.f;
        VecIO<H, VEC>::store(dst, z);
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + 4 * s, kpart = 0;
      if (nks > 1) { id = wid % ntiles; kpart = wid / ntiles; }
      if (id < ntiles) {
        const int mt = id / a.nt_total, nt = id - mt * a.nt_total;
        const H* ap = As + (mt * 32 + r) * LP + 8 * half + kpart * ksteps * 16;
        const H* bp = Bs + (nt * 32 + r) * LP + 8 * half + kpart * ksteps * 16;
        for (int ks = 0; ks < ksteps; ks++) {
          const hx8 af = *(const hx8*)(ap + ks * 16);
          const hx8 bf = *(const hx8*)(bp + ks * 16);
          acc[s] = mfma16<H>(af, bf, acc[s]);
        }
      }
    }
  }

#pragma unroll
  for (int s = 0; s < TPW; s++) {
    int id = wid + 4 * s;
    if (nks > 1) id = wid % ntiles;
    if (id < ntiles) {
      const int mt = id / a.nt_total, nt = id - mt * a.nt_total;
      const int ci = nt * 32 + r;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int co = co0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
        if (co < a.Cout && ci < a.Cin) atomicAdd(&a.dw[(long long)co * a.Cin + ci], acc[s][j]);
      }
    }
  }
}

template <typename H, int VEC, int TPW, bool XPRO, bool STRIDED>
static int pw_wgrad_bf16_launch(PwWgradArgs& a, hipStream_t st) {
  const int mt_total = ceil_div(a.Cout, 32);
  a.nt_total = ceil_div(a.Cin, 32);
  int g = (4 * TPW) / a.nt_total;
  if (g < 1) g = 1;
  if (g > mt_total) g = mt_total;
  a.mt_per_group = g;
  const int gy = ceil_div(mt_total, g);
  const long long steps_per_n = ceil_div_ll(a.P, 64);
  long long total = steps_per_n * a.N;
  int spb = (int)(total / 2048);
  if (spb < 4) spb = 4;
  if (spb > 64) spb = 64;
  if (spb > steps_per_n) spb = (int)steps_per_n;
  a.steps_per_block = spb;
  const long long gx = ceil_div_ll(steps_per_n, spb) * a.N;
  const size_t lds = (size_t)(a.mt_per_group + a.nt_total) * 32 * 72 * 2;
  X3D_DESCRIBE("pw_wgrad_bf16_kernel<%s, %d, %d, %d, %d>", HV<H>::name, VEC, TPW, (int)XPRO, (int)STRIDED);
  auto kern = pw_wgrad_bf16_kernel<H, VEC, TPW, XPRO, STRIDED>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_wgrad_bf16");
  return X3D_OK;
}

template <typename H, int VEC, bool XPRO, bool STRIDED>
static int pw_wgrad_bf16_tpw(PwWgradArgs& a, hipStream_t st) {
  const int nt = ceil_div(a.Cin, 32), mt = ceil_div(a.Cout, 32);
  const int tiles = nt * mt;
  if (tiles <= 4) return pw_wgrad_bf16_launch<H, VEC, 1, XPRO, STRIDED>(a, st);
  if (tiles <= 8) return pw_wgrad_bf16_launch<H, VEC, 2, XPRO, STRIDED>(a, st);
  if (tiles <= 16) return pw_wgrad_bf16_launch<H, VEC, 4, XPRO, STRIDED>(a, st);
  return pw_wgrad_bf16_launch<H, VEC, 8, XPRO, STRIDED>(a, st);
}

template <typename H>
static int pw_wgrad_bf16_dispatch(PwWgradArgs& a, int vec, bool xpro, hipStream_t st) {
  if (a.stride > 1) {
    if (xpro) { x3d_set_error("pw_wgrad: strided input takes no prologue"); return X3D_ERR_INVALID; }
    return pw_wgrad_bf16_tpw<H, 1, false, true>(a, st);
  }
  if (vec >= 8)
    return xpro ? pw_wgrad_bf16_tpw<H, 8, true, false>(a, st) : pw_wgrad_bf16_tpw<H, 8, false, false>(a, st);
  return xpro ? pw_wgrad_bf16_tpw<H, 1, true, false>(a, st) : pw_wgrad_bf16_tpw<H, 1, false, false>(a, st);
}

// ================================================================================================
// v2: the 16-byte-aligned fast path.
//  * 2-D decomposition of dW: a workgroup owns MG x NG tiles of 32x32 (MG*NG <= 8, blockIdx.y / .z), so
//    the fp32 partial it adds atomically at the end is <= 32 KB however wide the layer is, and the split
//    over points (blockIdx.x) can be made as fine as the machine needs.  The price is that the dY rows are
//    staged once per N-group and the X rows once per M-group (from L2 / Infinity Cache).
//  * 64-point steps are numbered across samples, so layers with few points per sample (P = 784) still
//    give every workgroup several steps to amortise its epilogue over.
//  * the next step's global loads are issued into registers before the current step's MFMAs.
//  * thread (tid>>3, tid&7) stages row (tid>>3) of every 32-row tile, points 8*(tid&7)..+7: one 16-byte
//    vector per tile row-block and tensor.
//  * strided shortcut (1x1x1, stride (1,2,2), valid): an output row segment of 8 points is the even
//    elements of 16 contiguous input elements -> two 16-byte loads, no scalar gather.
// ================================================================================================
//  * NTHR = 512 (round 2, the stage-5 layers 192 <-> 432): eight waves own up to 7 x 6 tiles, i.e. ALL of the narrower
//    operand and half of the wider one.  The run time of the 4-wave form follows the number of staged rows (4x2 tiles:
//    3552 rows, 100 us; 4x3: 2496 rows, 65 us); 7x6 / 6x7 stage 1248 / 1200.  Threads 256..511 stage the odd 32-row tiles.
//  * RAG: rows of P % 8 != 0 points (common.h, pw_ragged_rows) -- a separate instantiation, so that the element loads of a
//    row's last vector do not sit (as branches) in the loop of the aligned layers
template <typename H, int MG, int NG, bool XPRO, int STRIDED, int NTHR = 256, bool RAG = false>
__global__ __launch_bounds__(NTHR, (NTHR == 256 && MG * NG > 8 && MG * NG <= 12) ? 2 : 1) void pw_wgrad_bf16_v2_kernel(const PwWgradArgs a) {
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4; typedef typename HV<H>::x2 hx2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  // U sub-steps of 64 points are staged and multiplied per barrier pair: half the barriers and twice the loads in
  // flight per thread (a 64-point step is ~400 instructions between two barriers: the waves mostly wait)
  constexpr int BP = 64, U = NTHR == 512 ? 1 : 2, LP = U * BP + 8;   // (eight waves, 42 tiles: one sub-step's registers)
  constexpr int NW = NTHR / 64;                      // waves
  constexpr int NTS = NTHR / 256;                    // 32-row tiles staged side by side (thread group tsel takes tiles tsel, tsel + NTS, ...)
  constexpr int MGS = (MG + NTS - 1) / NTS, NGS = (NG + NTS - 1) / NTS;
  constexpr int TPW = (MG * NG + NW - 1) / NW;
  static_assert(NTHR == 256 || TPW > 1, "the split-K sharing of one or two tiles is written for four waves");
  H* As = (H*)smem_raw;              // [MG*32][LP]  dYraw
  H* Bs = As + MG * 32 * LP;            // [NG*32][LP]  f(X)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int srow = (tid >> 3) & 31, sp = (tid & 7) * 8;       // staging row within a 32-row tile, first point
  const int tsel = tid >> 8;                                  // which of the NTS side-by-side tiles this thread stages
  const int co0 = blockIdx.y * MG * 32, ci0 = blockIdx.z * NG * 32;
  const int steps_per_n = (int)((a.P + BP - 1) / BP);
  const int total_steps = steps_per_n * a.N;
  const int s_begin = blockIdx.x * a.steps_per_block;
  const int s_end = min(s_begin + a.steps_per_block, total_steps);
  const int mt_here = min(MG, (a.Cout - co0 + 31) / 32), nt_here = min(NG, (a.Cin - ci0 + 31) / 32);
  const int ntiles = mt_here * nt_here;
  const int nks = (TPW == 1 && ntiles <= 2) ? 4 / ntiles : 1;
  const int ksteps = (U * BP / 16) / nks;

  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

  hx8 rg[U][MGS], ry[U][MGS], rx[U][NGS], rx2[U][STRIDED ? NGS : 1];
  // per-row coefficients are loop invariants of this thread (rows co0 + i*32 + srow / ci0 + i*32 + srow): loaded
  // once here instead of from global inside every step's prologue (an exposed L2 round trip per 64-point step);
  // only the SE gate depends on the sample and is re-read when the step crosses into the next sample
  float cA[MGS], cB[MGS], cC[MGS], xs_[XPRO ? NGS : 1], xt_[XPRO ? NGS : 1], xg_[XPRO ? NGS : 1];
  const bool has_bn = a.coef || a.fold.sums;          // dY = A*g + B*yraw + C (coefficients from the table or from the sums)
#pragma unroll
  for (int i = 0; i < MGS; i++) {
    const int co = co0 + (i * NTS + tsel) * 32 + srow;
    const bool ok = has_bn && co < a.Cout && i * NTS + tsel < MG;
    cA[i] = 1.f; cB[i] = 0.f; cC[i] = 0.f;
    if (ok) bn_bwd_coef_load(a.coef, a.fold, co, false, cA[i], cB[i], cC[i]);
  }
  int n_gate = -1;
  auto load_gate = [&](int n) {
    if constexpr (XPRO) {
#pragma unroll
      for (int i = 0; i < NGS; i++) {
        const int ci = ci0 + (i * NTS + tsel) * 32 + srow;
        const bool inb = ci < a.Cin && i * NTS + tsel < NG;
        xs_[i] = inb ? a.xcoef[ci * 2] : 0.f; xt_[i] = inb ? a.xcoef[ci * 2 + 1] : 0.f;
        xg_[i] = (inb && a.xgate) ? a.xgate[(long long)n * a.Cin + ci] : 1.0f;
      }
    }
    n_gate = n;
  };
  // (n, stp) = (sample, 64-point step inside the sample) are carried incrementally: a division per step and lambda
  // was ~100 scalar instructions of the 380-860 in the loop
  auto issue = [&](int u, int n, int stp, bool live) {
    const long long p = live ? (long long)stp * BP + sp : a.P;   // a dead sub-step (past s_end) stages zeros
    hx8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (H)0.f;
#pragma unroll
    for (int i = 0; i < MGS; i++) {
      const int co = co0 + (i * NTS + tsel) * 32 + srow;
      rg[u][i] = z; ry[u][i] = z;
      if (i * NTS + tsel < MG && co < a.Cout && p < a.P) {
        const long long o = ((long long)n * a.Cout + co) * a.P + p;
        if (!RAG || a.P - p >= 8) {
          rg[u][i] = *(const hx8*)((const T*)a.g + o);
          if (has_bn) ry[u][i] = *(const hx8*)((const T*)a.yraw + o);
        } else {   // P % 8 != 0: the row ends inside this vector (zero fill: points are the reduction dimension)
          rg[u][i] = load8_ragged<T, hx8>((const T*)a.g + o, (int)(a.P - p));
          if (has_bn) ry[u][i] = load8_ragged<T, hx8>((const T*)a.yraw + o, (int)(a.P - p));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NGS; i++) {
      const int ci = ci0 + (i * NTS + tsel) * 32 + srow;
      rx[u][i] = z;
      if constexpr (STRIDED) rx2[u][i] = z;
      if (i * NTS + tsel < NG && ci < a.Cin && p < a.P) {
        if constexpr (STRIDED) {
          // strided shortcut: even input elements, STRIDED outputs per aligned load (common.h)
          strided_gather16<STRIDED>((const T*)a.x + ((long long)n * a.Cin + ci) * a.Pin, p, a.H, a.W, a.Ho, a.Wo, rx[u][i], rx2[u][i],
                                    RAG ? (int)min((long long)8, a.P - p) : 8);
        } else {
          const T* xsrc = (const T*)a.x + ((long long)n * a.Cin + ci) * a.Pin + p;
          if (!RAG || a.P - p >= 8) rx[u][i] = *(const hx8*)xsrc;
          else rx[u][i] = load8_ragged<T, hx8>(xsrc, (int)(a.P - p));
        }
      }
    }
  };
  auto commit = [&](int u, int n, int stp, bool live) {
#pragma unroll
    for (int i = 0; i < MGS; i++) {
      const int ti = i * NTS + tsel;
      if (ti >= MG) continue;
      const int co = co0 + ti * 32 + srow;
      H* dst = &As[(ti * 32 + srow) * LP + u * BP + sp];
      if (has_bn && co < a.Cout) {
        const float A = cA[i], B = cB[i], C = cC[i];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = A * (float)rg[u][i][e] + B * (float)ry[u][i][e] + C;
        const long long p = (long long)stp * BP + sp;
        if (p >= a.P || !live) {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = 0.f;      // C must not leak into padded points
        } else if (RAG && a.P - p < 8) {               // ... nor into the points past a ragged row end
#pragma unroll
          for (int e = 0; e < 8; e++) if (e >= (int)(a.P - p)) v[e] = 0.f;
        }
        VecIO<H, 8>::store(dst, v);
      } else {
        *(hx8*)dst = rg[u][i];
      }
    }
#pragma unroll
    for (int i = 0; i < NGS; i++) {
      const int ti = i * NTS + tsel;
      if (ti >= NG) continue;
      const int ci = ci0 + ti * 32 + srow;
      H* dst = &Bs[(ti * 32 + srow) * LP + u * BP + sp];
      if constexpr (STRIDED) {
        hx8 o;
#pragma unroll
        for (int e = 0; e < 4; e++) { o[e] = rx[u][i][2 * e]; o[4 + e] = rx2[u][i][2 * e]; }
        *(hx8*)dst = o;
      } else if constexpr (XPRO) {
        float v[8];
        if (live && n != n_gate) load_gate(n);
        const float s = xs_[i], t = xt_[i], g = xg_[i];
        const long long p = (long long)stp * BP + sp;
        const int nin = !live ? 0 : (RAG ? (int)min((long long)8, a.P - p) : (p < a.P ? 8 : 0));   // points of this vector inside the row
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (s * (float)rx[u][i][e] + t) * g;
        act_vec<8>(v, a.xact);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (e < nin) ? v[e] : 0.f;
        VecIO<H, 8>::store(dst, v);
      } else {
        *(hx8*)dst = rx[u][i];
      }
    }
  };

  // (sample, step-in-sample) of the sub-steps being consumed (c) and being loaded (i), advanced incrementally
  int n_c[U], stp_c[U], n_i[U], stp_i[U];
  {
    int n0 = s_begin / steps_per_n, st0 = s_begin - n0 * steps_per_n;
#pragma unroll
    for (int u = 0; u < U; u++) {
      n_c[u] = n0; stp_c[u] = st0;
      if (++st0 == steps_per_n) { st0 = 0; ++n0; }
    }
  }
  if (s_begin < s_end) {
    load_gate(n_c[0]);
#pragma unroll
    for (int u = 0; u < U; u++) issue(u, n_c[u], stp_c[u], s_begin + u < s_end);
  }
  for (int step = s_begin; step < s_end; step += U) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++) commit(u, n_c[u], stp_c[u], step + u < s_end);
    __syncthreads();
    {
      int n0 = n_c[U - 1], st0 = stp_c[U - 1];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (++st0 == steps_per_n) { st0 = 0; ++n0; }
        n_i[u] = n0; stp_i[u] = st0;
      }
    }
    if (step + U < s_end) {
#pragma unroll
      for (int u = 0; u < U; u++) issue(u, n_i[u], stp_i[u], step + U + u < s_end);
    }
#pragma unroll
    for (int u = 0; u < U; u++) { n_c[u] = n_i[u]; stp_c[u] = stp_i[u]; }
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + NW * s, kpart = 0;
      if (nks > 1) { id = wid % ntiles; kpart = wid / ntiles; }
      if (id < ntiles) {
        const int mt = id / nt_here, nt = id - mt * nt_here;
        const H* ap = As + (mt * 32 + r) * LP + 8 * half + kpart * ksteps * 16;
        const H* bp = Bs + (nt * 32 + r) * LP + 8 * half + kpart * ksteps * 16;
        for (int ks = 0; ks < ksteps; ks++) {
          const hx8 af = *(const hx8*)(ap + ks * 16);
          const hx8 bf = *(const hx8*)(bp + ks * 16);
          acc[s] = mfma16<H>(af, bf, acc[s]);
        }
      }
    }
  }

  // split-K waves (nks > 1: one or two tiles shared by the four waves) first add their partial tiles in LDS: the narrow
  // layers (24 x 24, 48 x 24 shortcuts) have thousands of workgroups adding into a few hundred dW addresses, all in one
  // or two L2 channels, and the atomics -- not the streaming -- were their bound
  if constexpr (TPW == 1) {
    if (nks > 1) {
      float* red = (float*)smem_raw;          // [4 waves][16][64] floats = 16 KB <= the staging tiles (2 * 32 * LP * 2 B = 17 KB)
      __syncthreads();                        // every wave is done with As / Bs
#pragma unroll
      for (int j = 0; j < 16; j++) red[(wid * 16 + j) * 64 + lane] = acc[0][j];
      __syncthreads();
      const int id = wid % ntiles, kpart = wid / ntiles;
      if (kpart == 0) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          float v = acc[0][j];
          for (int kp = 1; kp < nks; kp++) v += red[((id + kp * ntiles) * 16 + j) * 64 + lane];
          acc[0][j] = v;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < TPW; s++) {
    int id = wid + NW * s;
    bool writer = true;
    if (nks > 1) { id = wid % ntiles; writer = wid < ntiles; }
    if (id < ntiles && (s_begin < s_end || a.slab) && writer && !a.noflush) {
      const int mt = id / nt_here, nt = id - mt * nt_here;
      const int ci = ci0 + nt * 32 + r;
      // partial slab of this point chunk (plain stores; EVERY chunk writes its tiles, a chunk without steps zeros: the sum
      // over the slabs is taken by a later launch, x3d_hip.h dw_slab), or fp32 atomics into dw
      float* slab = a.slab ? a.slab + (long long)blockIdx.x * a.Cout * a.Cin : nullptr;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int co = co0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
        if (co < a.Cout && ci < a.Cin) {
          if (slab) slab[(long long)co * a.Cin + ci] = acc[s][j];
          else atomicAdd(&a.dw[(long long)co * a.Cin + ci], acc[s][j]);
        }
      }
    }
  }
}

template <typename H, int MG, int NG, bool XPRO, int STRIDED, int NTHR = 256, bool RAG = false>
static int pw_wgrad_v2_launch(PwWgradArgs& a, hipStream_t st) {
  if constexpr (!RAG) {
    if (a.ragged) return pw_wgrad_v2_launch<H, MG, NG, XPRO, STRIDED, NTHR, true>(a, st);
  }
  const int mt = ceil_div(a.Cout, 32), nt = ceil_div(a.Cin, 32);
  const int gy = ceil_div(mt, MG), gz = ceil_div(nt, NG);
  const long long total_steps = ceil_div_ll(a.P, 64) * a.N;
  X3D_REQUIRE(total_steps < (1ll << 31), "pw_wgrad: too many steps");
  const size_t lds = (size_t)(MG + NG) * 32 * ((NTHR == 512 ? 1 : 2) * 64 + 8) * 2;
  X3D_DESCRIBE("pw_wgrad_bf16_v2_kernel<%s, %d, %d, %d, %d, %d, %d>", HV<H>::name, MG, NG, (int)XPRO, STRIDED, NTHR, (int)RAG);
  // Partial-slab form (x3d_hip.h dw_slab) for the 12-tile groups -- the stage-5 layers 192 <-> 432, whose fp32 atomic flush is
  // 14-18 us of a 52-66 us launch (profiles/r05_noflush.txt).  Its grid must be known to the caller BEFORE the launch (and
  // without a device: dry plans), so it is sized from the two workgroups per CU the launch bounds promise, not from the
  // occupancy query.
  constexpr bool SLAB_FORM = STRIDED == 0 && NTHR == 256 && MG * NG == 12 && !RAG;
  if (x3d_parts_query && !SLAB_FORM) return X3D_OK;     // (query mode: no slab form here, the count stays 0)
  if (a.slab && !SLAB_FORM) { x3d_set_error("pw_wgrad: dw_slab given to a kernel without the slab form"); return X3D_ERR_INVALID; }
  auto kern = pw_wgrad_bf16_v2_kernel<H, MG, NG, XPRO, STRIDED, NTHR, RAG>;
  if (lds > 48 * 1024 && !x3d_parts_query) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_set = true;
    }
  }
  // one balanced round: as many workgroups as the chip holds at once (occupancy x CUs), the 64-point steps split
  // evenly among them.  A fixed steps-per-block left e.g. 1568 workgroups on 1280 slots: a second round at 22 %.
  static int slots = 0;
  if (slots == 0 && !(SLAB_FORM && (a.slab || x3d_parts_query))) {
    int nb = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NTHR, lds) != hipSuccess || nb < 1) nb = NTHR == 256 ? 2 : 1;
    slots = nb * cus;
  }
  const int slots_here = (SLAB_FORM && (a.slab || x3d_parts_query)) ? 2 * x3d_device_cus() : slots;
  long long gx_target = slots_here / (gy * gz);
  if (gx_target < 1) gx_target = 1;
  long long spb = ceil_div_ll(total_steps, gx_target);
  if (spb < 8) spb = 8;      // keeps the atomic partial (<= 32 KB) below ~10 % of the streamed bytes
  // every workgroup of a (y, z) tile group adds into the SAME dW tile: past ~500 workgroups per group the fp32 atomics on
  // a few hundred addresses are the bound (48 x 24 @ 28x28: 96 us with 1255 workgroups, 85 us with 392)
  if (ceil_div_ll(total_steps, spb) > 512) spb = ceil_div_ll(total_steps, 512);
  // 12-tile groups flush 12 K fp32 atomics per workgroup: 216 x 96 @ 14x14 with 242 workgroups per group 64 us, with
  // 157 (20 steps each) 58 us; the stage-5 layers (61 per group) are unaffected
  if (MG * NG > 8 && ceil_div_ll(total_steps, spb) > 160) spb = ceil_div_ll(total_steps, 160);
  const int spb_env = x3d_env_int("X3D_PW_WG_SPBMIN", 0);   // experiment hook
  if (spb < spb_env) spb = spb_env;
  a.steps_per_block = (int)spb;
  a.noflush = x3d_env_int("X3D_PW_WG_NOFLUSH", 0) == 1 ? 1 : 0;   // result-changing timing experiment (the atomic flush's share): -DX3D_EXPERIMENTS builds only
  long long gx = ceil_div_ll(total_steps, spb);
  if (gy * gz > 1 && xcd_pad_enabled()) gx = (gx + 7) & ~7ll;   // tile groups of one point chunk on one XCD (shared L2)
  if (x3d_parts_query) { *x3d_parts_query = (int)gx; return X3D_OK; }
  if (a.slab && gx != a.slab_parts) {
    x3d_set_error("pw_wgrad: dw_slab holds %d slabs, this launch writes %lld (x3d_pw_wgrad_dw_parts)", a.slab_parts, gx);
    return X3D_ERR_INVALID;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy, gz), dim3(NTHR), lds, st, a);
  X3D_LAUNCH_CHECK("pw_wgrad_bf16_v2");
  return X3D_OK;
}

template <typename H, bool XPRO, int STRIDED>
static int pw_wgrad_v2_pick(PwWgradArgs& a, hipStream_t st) {
  const int mt = ceil_div(a.Cout, 32), nt = ceil_div(a.Cin, 32);
  // X rows may carry the swish prologue: prefer few M-groups (each re-stages every X row of its N-group)
  const int MG = mt >= 3 ? 4 : mt, NG = nt >= 2 ? 2 : 1;
  if constexpr (STRIDED == 0) {
    // experiment hook X3D_PW_WG_WIDE=1, the stage-5 pair (192 <-> 432 channels: 6 x 14 / 14 x 6 tiles): eight-wave
    // workgroups that hold the whole narrow side and half of the wide one -- half the staged rows of the 12-tile groups.
    // Measured (tools/ab_wgrad5.sh, X3D_PW_WG_NOFLUSH=1 for the split): the streaming part gets faster (52 -> 38 us,
    // 37 -> 31 us) but one workgroup per CU means 104 point chunks instead of 56, and the fp32 atomic flush -- 35 MB
    // instead of 19 MB -- grows from 14 to 25-28 us: 64 vs 66 us and 59 vs 52 us in total, so the 12-tile groups stay.
    if (x3d_env_int("X3D_PW_WG_WIDE", 0) == 1) {
      if (nt >= 4 && nt <= 6 && mt >= 10 && mt <= 14) return pw_wgrad_v2_launch<H, 7, 6, XPRO, STRIDED, 512>(a, st);
      if (mt >= 4 && mt <= 6 && nt >= 10 && nt <= 14) return pw_wgrad_v2_launch<H, 6, 7, XPRO, STRIDED, 512>(a, st);
    }
  }
  if constexpr (STRIDED == 0) {
    // experiment hook X3D_PW_WG_NG4=1: 128 x 128 tiles for the wide layers (stage 5: 192 x 432) halve the re-reads of
    // every dY / X row by the other tile groups, but need 232-252 VGPRs + 64 AGPRs (one workgroup per CU):
    // measured slower (80 -> 100 us), so 128 x 64 stays the default
    if (MG == 4 && nt >= 4 && x3d_env_int("X3D_PW_WG_NG4", 0) == 1) return pw_wgrad_v2_launch<H, 4, 4, XPRO, STRIDED>(a, st);
  }
  if constexpr (STRIDED == 0) {
    // wide layers: every dY (+ yraw) row is staged once per N-group and every X row once per M-group, so the tile shape
    // sets the traffic: gz * (1 or 2) * Cout + gy * Cin rows of P points.  12-tile shapes (3 tiles per wave) still
    // fit two workgroups per CU; pick the cheapest of 4x2 / 4x3 / 3x4 (216 x 96: 1056 -> 624 rows, 192 x 432:
    // 3552 -> 2400, 432 x 192: 3360 -> 2496).
    if (MG == 4 && nt >= 3 && x3d_env_int("X3D_PW_WG_T12", 1) != 0) {   // A/B switch: 0 = 4x2 only
      const int dyr = (a.coef || a.fold.sums) ? 2 : 1;
      auto rows = [&](int mg, int ng) { return (long long)ceil_div(nt, ng) * dyr * a.Cout + (long long)ceil_div(mt, mg) * a.Cin; };
      const long long r42 = rows(4, 2), r43 = rows(4, 3), r34 = rows(3, 4);
      if (r43 < r42 && r43 <= r34) return pw_wgrad_v2_launch<H, 4, 3, XPRO, STRIDED>(a, st);
      if (r34 < r42) return pw_wgrad_v2_launch<H, 3, 4, XPRO, STRIDED>(a, st);
    }
  }
  if (MG == 1) return NG == 1 ? pw_wgrad_v2_launch<H, 1, 1, XPRO, STRIDED>(a, st) : pw_wgrad_v2_launch<H, 1, 2, XPRO, STRIDED>(a, st);
  if (MG == 2) return NG == 1 ? pw_wgrad_v2_launch<H, 2, 1, XPRO, STRIDED>(a, st) : pw_wgrad_v2_launch<H, 2, 2, XPRO, STRIDED>(a, st);
  return NG == 1 ? pw_wgrad_v2_launch<H, 4, 1, XPRO, STRIDED>(a, st) : pw_wgrad_v2_launch<H, 4, 2, XPRO, STRIDED>(a, st);
}

// returns -1 when the fast path does not apply (caller falls back to the generic kernel)
template <typename H>
static int pw_wgrad_v2_dispatch(PwWgradArgs& a, int vec, bool xpro, hipStream_t st) {
  if (vec < 8 && !a.ragged) return -1;   // (ragged: P % 8 != 0 with 16-byte aligned tensors, set by the entry point)
  if (a.stride > 1) {
    if (a.stride != 2 || xpro) return -1;
    switch (strided_gather_gv(a.W, a.Wo, a.P, a.x)) {
      case 4: return pw_wgrad_v2_pick<H, false, 4>(a, st);
      case 2: return pw_wgrad_v2_pick<H, false, 2>(a, st);
      case 1: return pw_wgrad_v2_pick<H, false, 1>(a, st);
      default: return -1;
    }
  }
  return xpro ? pw_wgrad_v2_pick<H, true, 0>(a, st) : pw_wgrad_v2_pick<H, false, 0>(a, st);
}
