// Exact-fp32 pointwise weight gradient, the operand streams PIPELINED (fp32 storage: BASELINE config 2) -- pw_wgrad_f32r.h's
// tile groups, run lengths, slab form and arithmetic with pw_gemm_f32p.h's memory pipeline:
//   * every load is an UNCONDITIONAL buffer load (rows past the channel count, columns past P, steps past the run take the
//     out-of-range offset: zeros, no traffic) -- pw_wgrad_f32r.h's sit behind run-time guards, one step in flight, each commit
//     waiting with vmcnt(0) (a stage-4 launch: 11 steps of ~2.5 us of latency each) -- with D steps in flight in D register sets,
//     the step loop unrolled by D;
//   * the thread-invariant part of every address is a scalar offset, rows of P % 4 != 0 points load their last, partial vector
//     4 - P % 4 elements early and rotate it (no load crosses a row end);
//   * the tile groups of one point chunk read the same rows: they take consecutive slots of ONE XCD (workgroup ids = xcd mod 8).
// Same products, same fp32 sums per partial tile, same partition of the points between workgroups.
#pragma once
#include "pw_wgrad_f32r.h"

#ifdef WGP_STAMPS      // in-kernel stamps of workgroup 0 (tools/f32p_stamps.py wgrad): s_memtime at the phase boundaries
__device__ unsigned long long wgp_stamps[64];
#define WGP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && (i) < 64) wgp_stamps[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int x3d_debug_wgp_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(wgp_stamps), sizeof(wgp_stamps)) == hipSuccess ? 0 : 1; }
#else
#define WGP_STAMP(i) do { } while (0)
#endif
constexpr int WGP_OOB = 0x7fffff00;   // buffer offset past every tensor (host check): loads return 0
typedef __attribute__((ext_vector_type(4))) unsigned int wgp_u32x4;

template <int MTG, int NTG, bool XPRO, bool RAG>
__global__ __launch_bounds__(256) void pw_wgrad_f32p_kernel(const PwWgradRArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BP = 32, LP = 33;
  constexpr int RA = MTG * 32, RB = NTG * 32, NTILE = MTG * NTG, TPW = (NTILE + 3) / 4;
  constexpr int NVA = RA * 8 / 256, NVB = RB * 8 / 256;        // float4 staging vectors per thread and step (rows x 8 vectors)
  constexpr int D = (2 * NVA + NVB) * 4 * 3 + TPW * 16 <= 112 ? 3 : 2;   // steps in flight (register sets)
  static_assert(RA * 8 % 256 == 0 && RB * 8 % 256 == 0, "row counts must fill the workgroup");
  float* As = smem;                          // [2][RA][LP]
  float* Bs = As + 2 * RA * LP;              // [2][RB][LP]
  float* Ca = Bs + 2 * RB * LP;              // [RA][4]
  float* Cb = Ca + RA * 4;                   // [RB][4]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int groups = a.mgroups * a.ngroups;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = slot % groups, part = (slot / groups) * 8 + xcd;      // part: the (sample, point chunk) -- the slab index
  const int mg = grp / a.ngroups, ng = grp - mg * a.ngroups;
  const int co0 = mg * RA, ci0 = ng * RB;
  const int P = (int)a.P;
  const int steps_per_n = (P + BP - 1) / BP;
  const int chunks_per_n = (steps_per_n + a.steps_per_block - 1) / a.steps_per_block;
  if (part >= chunks_per_n * a.N) return;
  const int n = part / chunks_per_n;
  const int chunk = part - n * chunks_per_n;
  const int s_begin = chunk * a.steps_per_block;
  const int s_end = min(s_begin + a.steps_per_block, steps_per_n);
  if (s_begin >= s_end) return;
  WGP_STAMP(0);
  const bool two = a.coef || a.fold.sums;    // dY = A g + B yraw + C (else g as it is)

  for (int row = tid; row < RA; row += 256) {
    const int co = co0 + row;
    float cA = 1.f, cB = 0.f, cC = 0.f;
    if (co < a.Cout && two) bn_bwd_coef_load(a.coef, a.fold, co, false, cA, cB, cC);
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

  const int gbytes = a.N * a.Cout * P * 4, xbytes = a.N * a.Cin * P * 4;
  const __amdgpu_buffer_rsrc_t rgr = __builtin_amdgcn_make_buffer_rsrc((float*)a.g, 0, gbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((float*)(two ? a.yraw : a.g), 0, two ? gbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc((float*)a.x, 0, xbytes, 0x00020000);

  // staging role: rows row0 + 32 i of either operand, the 4 points from pv * 4 of a step
  const int row0 = tid >> 3, pv = tid & 7;
  const int q4 = P & 3;
  const int vrow = (row0 * P + pv * 4) * 4;
  const int abase = (n * a.Cout + co0) * P * 4, bbase = (n * a.Cin + ci0) * P * 4;      // scalar
  const int alim = a.Cout - co0, blim = a.Cin - ci0;                                   // rows of the group inside the tensor

  wgp_u32x4 rg[D][NVA], ry[D][NVA], rx[D][NVB];
  auto issue = [&](auto SET, int step) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int p0 = step * BP;
    const int pcol = p0 + pv * 4;
    const bool colok = step < s_end && pcol < P;
    // (the early start of a partial vector may reach in front of the step: with row 0 / vector 0 the per-thread offset would go
    // negative -- 16 bytes move from the scalar to the per-thread offset; at p0 = 0 the partial vector is never vector 0, P >= 4)
    const int sb = (RAG && p0 > 0) ? 16 : 0;
    int vo = vrow + sb;
    if constexpr (RAG) { if (pcol + 4 > P) vo -= (4 - q4) * 4; }
#pragma unroll
    for (int i = 0; i < NVA; i++) {
      const int v = (colok && row0 + 32 * i < alim) ? vo : WGP_OOB;
      const int so = abase + (32 * i * P + p0) * 4 - sb;
      rg[S][i] = __builtin_bit_cast(wgp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rgr, v, so, 0));
      ry[S][i] = __builtin_bit_cast(wgp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(ryr, v, so, 0));
    }
#pragma unroll
    for (int i = 0; i < NVB; i++) {
      const int v = (colok && row0 + 32 * i < blim) ? vo : WGP_OOB;
      rx[S][i] = __builtin_bit_cast(wgp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rxr, v, bbase + (32 * i * P + p0) * 4 - sb, 0));
    }
  };
  const bool act_swish = a.xact == X3D_ACT_SWISH;
  const float act_floor = a.xact == X3D_ACT_RELU ? 0.f : -INFINITY;
  auto commit = [&](auto SET, int step, int buf) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int pcol = step * BP + pv * 4;
    const bool part_ = RAG && pcol + 4 > P;
    const bool sh1 = part_ && q4 == 1, sh2 = part_ && q4 == 2, sh3 = part_ && q4 == 3;
    auto elem = [&](const wgp_u32x4& v, int e) __attribute__((always_inline)) -> float {
      float x = __uint_as_float(v[e]);
      if constexpr (RAG) {
        if (e + 1 < 4) x = sh3 ? __uint_as_float(v[(e + 1) & 3]) : x;
        if (e + 2 < 4) x = sh2 ? __uint_as_float(v[(e + 2) & 3]) : x;
        if (e + 3 < 4) x = sh1 ? __uint_as_float(v[(e + 3) & 3]) : x;
      }
      return x;
    };
    float* A_ = As + buf * RA * LP;
    float* B_ = Bs + buf * RB * LP;
    // dY rows: whatever lands in columns past P (the table's C, a ragged row's rotated leftovers) meets an exact zero on the x side
#pragma unroll
    for (int i = 0; i < NVA; i++) {
      const int row = row0 + 32 * i;
      const f32x4 c = *(const f32x4*)(Ca + row * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) A_[row * LP + pv * 4 + e] = c[0] * elem(rg[S][i], e) + c[1] * elem(ry[S][i], e) + c[2];
    }
#pragma unroll
    for (int i = 0; i < NVB; i++) {
      const int row = row0 + 32 * i;
      float val[4];
#pragma unroll
      for (int e = 0; e < 4; e++) val[e] = elem(rx[S][i], e);
      if constexpr (XPRO) {
        const f32x4 c = *(const f32x4*)(Cb + row * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float u = (c[0] * val[e] + c[1]) * c[2];
          const float sw = swishf_(u), mx = fmaxf(u, act_floor);
          val[e] = act_swish ? sw : mx;
        }
      }
      if constexpr (XPRO || RAG) {           // (without a prologue and with whole vectors the out-of-range loads are the zeros)
#pragma unroll
        for (int e = 0; e < 4; e++) val[e] = (row < blim && step < s_end && pcol + e < P) ? val[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) B_[row * LP + pv * 4 + e] = val[e];
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

  int par = 0;
  auto sub = [&](auto SET, int step) __attribute__((always_inline)) {
    const int buf = par;
    par ^= 1;
    commit(SET, step, buf);
    __syncthreads();      // step visible; every wave is past the MFMAs that read the other buffer
    issue(SET, step + D);
    const float* A_ = As + buf * RA * LP;
    const float* B_ = Bs + buf * RB * LP;
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + 4 * s;
      if (NTILE % 4 == 0 || id < NTILE) {
        const int mt = id / NTG, nt = id - mt * NTG;
        const float* ap = A_ + (mt * 32 + r) * LP + half;
        const float* bp = B_ + (nt * 32 + r) * LP + half;
        float av[BP / 2], bv[BP / 2];
#pragma unroll
        for (int i = 0; i < BP / 2; i++) { av[i] = ap[2 * i]; bv[i] = bp[2 * i]; }
#pragma unroll
        for (int i = 0; i < BP / 2; i++) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc[s], 0, 0, 0);
      }
    }
  };

  issue(std::integral_constant<int, 0>(), s_begin);
  issue(std::integral_constant<int, 1>(), s_begin + 1);
  if constexpr (D == 3) issue(std::integral_constant<int, 2>(), s_begin + 2);
  WGP_STAMP(1);
  __syncthreads();        // tables in place
  WGP_STAMP(2);
  // (no early exit inside the unrolled body -- pw_gemm_f32p.h: up to D - 1 sub-steps past the run multiply zeros)
  for (int step = s_begin; step < s_end; step += D) {
    WGP_STAMP(3 + (step - s_begin) / D);
    sub(std::integral_constant<int, 0>(), step);
    sub(std::integral_constant<int, 1>(), step + 1);
    if constexpr (D == 3) sub(std::integral_constant<int, 2>(), step + 2);
  }

  WGP_STAMP(60);
#pragma unroll
  for (int s = 0; s < TPW; s++) {
    const int id = wid + 4 * s;
    if (NTILE % 4 == 0 || id < NTILE) {
      const int mt = id / NTG, nt = id - mt * NTG;
      const int ci = ci0 + nt * 32 + r;
      // partial slab of this (sample, point chunk) -- every part has steps, so every slab is written whole -- or fp32 atomics
      float* slab = a.slab ? a.slab + (long long)part * a.Cout * a.Cin : nullptr;
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
  WGP_STAMP(61);
}

template <int MTG, int NTG, bool XPRO, bool RAG>
static int wgrad_f32p_launch(PwWgradRArgs& a, hipStream_t st) {
  X3D_DESCRIBE("pw_wgrad_f32p_kernel<%d, %d, %d, %d>", MTG, NTG, (int)XPRO, (int)RAG);
  a.mgroups = ceil_div(ceil_div(a.Cout, 32), MTG);
  a.ngroups = ceil_div(ceil_div(a.Cin, 32), NTG);
  const size_t lds = ((size_t)2 * (MTG + NTG) * 32 * 33 + (size_t)(MTG + NTG) * 32 * 4) * sizeof(float);
  auto kern = pw_wgrad_f32p_kernel<MTG, NTG, XPRO, RAG>;
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
  const long long parts = ceil_div_ll(steps_per_n, spb) * a.N;
  if (x3d_parts_query) { *x3d_parts_query = (int)parts; return X3D_OK; }     // (x3d_pw_wgrad_dw_parts: one slab per part)
  if (a.slab && parts != a.slab_parts) {
    x3d_set_error("pw_wgrad: dw_slab holds %d slabs, this launch writes %lld (x3d_pw_wgrad_dw_parts)", a.slab_parts, parts);
    return X3D_ERR_INVALID;
  }
  const long long grid = ceil_div_ll(parts, 8) * 8 * groups;       // (parts padded to the 8 XCDs: a workgroup past the last part returns at once)
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_wgrad_f32p");
  return X3D_OK;
}

// tile group of a layer: pw_wgrad_f32r.h's choice.  -1: not covered (the caller goes on to pw_wgrad_f32r.h)
template <bool XPRO>
static int wgrad_f32p_pick(PwWgradRArgs& a, hipStream_t st) {
  if (x3d_env_int("X3D_PW_F32P", 1) == 0) return -1;
  const long long big = ((long long)a.N * (a.Cout > a.Cin ? a.Cout : a.Cin) + 256) * a.P * 4;
  if (big >= (long long)WGP_OOB || a.P < 4) return -1;             // 32-bit buffer offsets
  if ((((uintptr_t)a.g | (uintptr_t)a.yraw | (uintptr_t)a.x) & 3) != 0) return -1;
  const int mt = ceil_div(a.Cout, 32), nt = ceil_div(a.Cin, 32);
  int ntg = nt >= 4 ? 4 : nt;
  if (nt == 5 || nt == 6) ntg = 3;                                  // 5 -> 3 + 2, 6 -> 3 + 3 (less padding than 4 + 1 / 4 + 2)
  int mtg = 8 / ntg;
  if (mtg > mt) mtg = mt;
  if (mtg == 3) mtg = 2;
  if (mtg > 4 && mtg < 8) mtg = 4;
  const bool rag = (a.P & 3) != 0;
#define WGP_CASE(M_, N_) if (mtg == M_ && ntg == N_) return rag ? wgrad_f32p_launch<M_, N_, XPRO, true>(a, st) : wgrad_f32p_launch<M_, N_, XPRO, false>(a, st);
  WGP_CASE(1, 1) WGP_CASE(2, 1) WGP_CASE(4, 1) WGP_CASE(8, 1) WGP_CASE(1, 2) WGP_CASE(2, 2) WGP_CASE(4, 2) WGP_CASE(1, 3) WGP_CASE(2, 3)
  WGP_CASE(1, 4) WGP_CASE(2, 4)
#undef WGP_CASE
  return -1;
}
