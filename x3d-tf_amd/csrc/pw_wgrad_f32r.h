// Exact-fp32 pointwise weight gradient, restructured like pw_gemm_f32r.h (fp32 storage: BASELINE config 2).
//
// pw_wgrad_kernel<float> stages a [rows][32 points] block of BOTH operands synchronously per 32-point step (stage -> barrier ->
// MFMA -> barrier), every workgroup owning up to 32 output tiles and only 4 steps: thousands of workgroups, each ending in
// tens of thousands of fp32 atomics on the same few thousand addresses (96 x 216 on 13x10x10: 100 us for 1.7 GFLOP).  Here
//   * a workgroup owns a GROUP of at most 8 output tiles (MTG x NTG 32x32 tiles, one or two per wave) and a LONG run of
//     steps: the launch has ~2 workgroups per CU in all, so the atomic flush is (number of point chunks) x Cout x Cin
//     with 5-10x fewer point chunks; the operand rows a group needs are re-read by the other groups of the same chunk from L2;
//   * steps are double-buffered in LDS, the next step's vectors are loaded into registers while the current one is multiplied
//     (one barrier per step), the BN-backward / prologue coefficients sit in LDS tables.
// Same products, same fp32 sums per partial tile; only the partition of the points between workgroups changes.
#pragma once
#include "common.h"

struct PwWgradRArgs {
  const void* g; const void* yraw; const float* coef;
  BnBwdFold fold;            // sums != NULL: the dY coefficients are derived from the BatchNorm-backward sums (x3d_hip.h coef_fold; never published here)
  const void* x; const float* xcoef; const float* xgate; int xact;
  float* dw;
  float* slab;               // NULL | partial weight gradients [gridDim.x][Cout][Cin], plain stores (x3d_hip.h dw_slab)
  int slab_parts;            // slabs the buffer holds (host side: checked against the grid)
  int N, Cout, Cin;
  long long P;
  int mgroups, ngroups;      // tile groups along Cout / Cin
  int steps_per_block;
};

template <int MTG, int NTG, bool XPRO>
__global__ __launch_bounds__(256) void pw_wgrad_f32r_kernel(const PwWgradRArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BP = 32, LP = 33;
  constexpr int RA = MTG * 32, RB = NTG * 32, NTILE = MTG * NTG, TPW = (NTILE + 3) / 4;
  constexpr int NVA = RA * 8 / 256, NVB = RB * 8 / 256;        // float4 staging vectors per thread and step (rows x 8 vectors)
  static_assert(RA * 8 % 256 == 0 && RB * 8 % 256 == 0, "row counts must fill the workgroup");
  float* As = smem;                          // [2][RA][LP]
  float* Bs = As + 2 * RA * LP;              // [2][RB][LP]
  float* Ca = Bs + 2 * RB * LP;              // [RA][4]
  float* Cb = Ca + RA * 4;                   // [RB][4]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int grp = blockIdx.y;
  const int mg = grp / a.ngroups, ng = grp - mg * a.ngroups;
  const int co0 = mg * RA, ci0 = ng * RB;
  const int steps_per_n = (int)((a.P + BP - 1) / BP);
  const int chunks_per_n = (steps_per_n + a.steps_per_block - 1) / a.steps_per_block;
  const int n = blockIdx.x / chunks_per_n;
  const int chunk = blockIdx.x - n * chunks_per_n;
  const int s_begin = chunk * a.steps_per_block;
  const int s_end = min(s_begin + a.steps_per_block, steps_per_n);
  if (s_begin >= s_end) return;

  for (int row = tid; row < RA; row += 256) {
    const int co = co0 + row;
    const bool ok = co < a.Cout && (a.coef || a.fold.sums);
    float cA = 1.f, cB = 0.f, cC = 0.f;
    if (ok) bn_bwd_coef_load(a.coef, a.fold, co, false, cA, cB, cC);
    Ca[row * 4] = cA; Ca[row * 4 + 1] = cB; Ca[row * 4 + 2] = cC;
  }
  if constexpr (XPRO) {
    for (int row = tid; row < RB; row += 256) {
      const int ci = ci0 + row;
      const bool ok = ci < a.Cin;
      Cb[row * 4] = ok ? a.xcoef[ci * 2] : 0.f; Cb[row * 4 + 1] = ok ? a.xcoef[ci * 2 + 1] : 0.f;
      Cb[row * 4 + 2] = (ok && a.xgate) ? a.xgate[(long long)n * a.Cin + ci] : 1.0f;
    }
  }

  f32x4 rg[NVA], ry[NVA], rx[NVB];
  auto load4 = [&](const float* base, long long o, long long p, f32x4& v) __attribute__((always_inline)) {
    v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p + 4 <= a.P) v = *(const f32x4*)(base + o);        // (rows of P % 4 != 0 points: unaligned 16-byte loads, element tails)
    else {
#pragma unroll
      for (int e = 0; e < 4; e++) if (p + e < a.P) v[e] = base[o + e];
    }
  };
  auto issue = [&](int step) __attribute__((always_inline)) {
    const long long p0 = (long long)step * BP;
#pragma unroll
    for (int i = 0; i < NVA; i++) {
      const int v = tid + i * 256, row = v >> 3, pv = v & 7;
      const int co = co0 + row;
      const long long p = p0 + pv * 4;
      rg[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ry[i] = rg[i];
      if (co < a.Cout && p < a.P) {
        const long long o = ((long long)n * a.Cout + co) * a.P + p;
        load4((const float*)a.g, o, p, rg[i]);
        if (a.coef || a.fold.sums) load4((const float*)a.yraw, o, p, ry[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NVB; i++) {
      const int v = tid + i * 256, row = v >> 3, pv = v & 7;
      const int ci = ci0 + row;
      const long long p = p0 + pv * 4;
      rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ci < a.Cin && p < a.P) load4((const float*)a.x, ((long long)n * a.Cin + ci) * a.P + p, p, rx[i]);
    }
  };
  auto commit = [&](int step, int buf) __attribute__((always_inline)) {
    const long long p0 = (long long)step * BP;
    float* A_ = As + buf * RA * LP;
    float* B_ = Bs + buf * RB * LP;
#pragma unroll
    for (int i = 0; i < NVA; i++) {
      const int v = tid + i * 256, row = v >> 3, pv = v & 7;
      const long long p = p0 + pv * 4;
      const bool rok = co0 + row < a.Cout;
      const float cA = Ca[row * 4], cB = Ca[row * 4 + 1], cC = Ca[row * 4 + 2];
#pragma unroll
      for (int e = 0; e < 4; e++) A_[row * LP + pv * 4 + e] = (rok && p + e < a.P) ? (cA * rg[i][e] + cB * ry[i][e] + cC) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NVB; i++) {
      const int v = tid + i * 256, row = v >> 3, pv = v & 7;
      const long long p = p0 + pv * 4;
      const bool rok = ci0 + row < a.Cin;
      float val[4];
#pragma unroll
      for (int e = 0; e < 4; e++) val[e] = rx[i][e];
      if constexpr (XPRO) {
        const float s_ = Cb[row * 4], t_ = Cb[row * 4 + 1], g_ = Cb[row * 4 + 2];
#pragma unroll
        for (int e = 0; e < 4; e++) val[e] = (s_ * val[e] + t_) * g_;
        act_vec<4>(val, a.xact);
      }
#pragma unroll
      for (int e = 0; e < 4; e++) B_[row * LP + pv * 4 + e] = (rok && p + e < a.P) ? val[e] : 0.f;
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

  issue(s_begin);
  __syncthreads();        // tables in place
  for (int step = s_begin; step < s_end; ++step) {
    const int buf = (step - s_begin) & 1;
    commit(step, buf);
    __syncthreads();      // step visible; every wave is past the MFMAs that read the other buffer
    if (step + 1 < s_end) issue(step + 1);
    const float* A_ = As + buf * RA * LP;
    const float* B_ = Bs + buf * RB * LP;
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + 4 * s;
      if (id < NTILE) {
        const int mt = id / NTG, nt = id - mt * NTG;
        const float* ap = A_ + (mt * 32 + r) * LP + half;
        const float* bp = B_ + (nt * 32 + r) * LP + half;
#pragma unroll
        for (int kk = 0; kk < BP; kk += 2) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc[s], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int s = 0; s < TPW; s++) {
    const int id = wid + 4 * s;
    if (id < NTILE) {
      const int mt = id / NTG, nt = id - mt * NTG;
      const int ci = ci0 + nt * 32 + r;
      // partial slab of this (sample, point chunk) -- every workgroup of the grid has steps, so every slab is written whole --
      // or fp32 atomics into dw (x3d_hip.h dw_slab; the flush of 32-64 chunks x [Cout][Cin] is 10-15 us of a ~55 us launch)
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

// X3D_PW_F32R=0: A/B hook shared with pw_gemm_f32r.h
template <int MTG, int NTG, bool XPRO>
static int wgrad_f32r_launch(PwWgradRArgs& a, hipStream_t st) {
  X3D_DESCRIBE("pw_wgrad_f32r_kernel<%d, %d, %d>", MTG, NTG, (int)XPRO);
  a.mgroups = ceil_div(ceil_div(a.Cout, 32), MTG);
  a.ngroups = ceil_div(ceil_div(a.Cin, 32), NTG);
  const size_t lds = ((size_t)2 * (MTG + NTG) * 32 * 33 + (size_t)(MTG + NTG) * 32 * 4) * sizeof(float);
  auto kern = pw_wgrad_f32r_kernel<MTG, NTG, XPRO>;
  static bool attr_set = false;
  if (!attr_set && !x3d_parts_query) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  const int cus = x3d_device_cus();
  const int groups = a.mgroups * a.ngroups;
  const long long steps_per_n = ceil_div_ll(a.P, 32);
  // about two workgroups per CU in all: few point chunks = few atomic flushes, long runs = the pipeline's latency amortised
  long long chunks = (2ll * cus) / groups;
  if (chunks < a.N) chunks = a.N;                                  // (a chunk does not cross samples)
  long long per_n = chunks / a.N;
  if (per_n < 1) per_n = 1;
  long long spb = ceil_div_ll(steps_per_n, per_n);
  if (spb < 4) spb = 4;
  if (spb > steps_per_n) spb = steps_per_n;
  a.steps_per_block = (int)spb;
  const long long gx = ceil_div_ll(steps_per_n, spb) * a.N;
  if (x3d_parts_query) { *x3d_parts_query = (int)gx; return X3D_OK; }     // (x3d_pw_wgrad_dw_parts: one slab per blockIdx.x)
  if (a.slab && gx != a.slab_parts) {
    x3d_set_error("pw_wgrad: dw_slab holds %d slabs, this launch writes %lld (x3d_pw_wgrad_dw_parts)", a.slab_parts, gx);
    return X3D_ERR_INVALID;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)groups), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_wgrad_f32r");
  return X3D_OK;
}

// tile group of a layer: at most 8 tiles (two per wave); wide inputs take four column tiles
template <bool XPRO>
static int wgrad_f32r_pick(PwWgradRArgs& a, hipStream_t st) {
  const int mt = ceil_div(a.Cout, 32), nt = ceil_div(a.Cin, 32);
  int ntg = nt >= 4 ? 4 : nt;
  if (nt == 5 || nt == 6) ntg = 3;                                  // 5 -> 3 + 2, 6 -> 3 + 3 (less padding than 4 + 1 / 4 + 2)
  int mtg = 8 / ntg;
  if (mtg > mt) mtg = mt;
  if (mtg == 3) mtg = 2;
  if (mtg > 4 && mtg < 8) mtg = 4;
#define WG_CASE(M_, N_) if (mtg == M_ && ntg == N_) return wgrad_f32r_launch<M_, N_, XPRO>(a, st);
  WG_CASE(1, 1) WG_CASE(2, 1) WG_CASE(4, 1) WG_CASE(8, 1) WG_CASE(1, 2) WG_CASE(2, 2) WG_CASE(4, 2) WG_CASE(1, 3) WG_CASE(2, 3)
  WG_CASE(1, 4) WG_CASE(2, 4)
#undef WG_CASE
  return -1;
}
