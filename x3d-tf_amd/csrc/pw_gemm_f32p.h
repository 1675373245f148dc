// Exact-fp32 pointwise GEMM, weights resident in LDS, the activation stream PIPELINED (fp32 storage: BASELINE config 2).
//
// pw_gemm_f32r.h's kernel keeps ONE activation chunk in flight (loaded under the MFMAs of the chunk before it, behind
// run-time guards: every load sits in its own branch and the commit waits with vmcnt(0)) and rebuilds the tile / row / column
// indices, with two integer divisions, at every issue and commit.  Stage-removal runs (profiles/r06_ab_f32r_parts.txt): of
// the 56 us of 96 -> 216 on 13x10x10 the MFMAs are 12, the statistics atomics 12, and 432 -> 192 on 13x5x5 (14 chunks a tile,
// one workgroup per CU) spends 46 of 97 us waiting for its loads.  Here
//   * every load is an UNCONDITIONAL buffer load (rows past K, columns past P and steps past the workgroup's last tile take
//     the out-of-range offset: zeros, no traffic), so vmcnt counts exactly and D register sets -- D chunks -- stay in flight;
//     the step loop is unrolled by D, the set index is a compile-time constant;
//   * (tile, chunk) is a flat step counter with two cursors (loads run D steps ahead of the MFMAs), advanced by additions;
//     the thread-invariant part of every address is a scalar offset;
//   * the SE gate of a row is loaded with the row (no per-sample table refill between barriers);
//   * the epilogue stores / loads are buffer operations with a per-lane base and scalar row offsets;
//   * the statistics of the NT waves of a row block are added in LDS: one pair of atomics per channel and workgroup; the sums
//     over the 32 lanes of a half wave are DPP adds (as shuffles -- ds_bpermute round trips, each waited for -- the 32 sums were
//     18,000 of the 64,000 cycles of a workgroup: in-kernel stamps, profiles/r06_f32p_stamps.txt);
//   * the row groups of one tile range take consecutive slots of one XCD (its L2 serves all but the first read of a tile).
// Arithmetic, operand order and the k order inside an output element are pw_gemm_f32r.h's (and pw_gemm.h's): same results up
// to the order of the statistics atomics.
#pragma once
#include "pw_gemm_f32r.h"

#ifndef F32P_EXP
#define F32P_EXP 0    // timing experiments (tools/ab_f32r_parts.sh; results WRONG unless 0): 1 no MFMA, 2 no weight load, 4 no epilogue,
#endif                // 8 no activation traffic (every load out of range), 16 no statistics flush, 32 no commit, 64 no barrier in the step
#if (F32P_EXP & 256)      // in-kernel stamps of workgroup 0 / the last workgroup (tools/f32p_stamps.py): s_memtime at the phase boundaries
__device__ unsigned long long f32p_stamps[2][64];
#define F32P_STAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (i) < 64) f32p_stamps[blockIdx.x == 0 ? 0 : 1][i] = __builtin_readcyclecounter(); } while (0)
extern "C" int x3d_debug_f32p_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(f32p_stamps), sizeof(f32p_stamps)) == hipSuccess ? 0 : 1; }
#else
#define F32P_STAMP(i) do { } while (0)
#endif
constexpr int F32P_OOB = 0x7fffff00;   // buffer offset past every tensor (host check): loads return 0, stores are dropped
typedef __attribute__((ext_vector_type(4))) unsigned int f32p_u32x4;

template <int MT, int NT, int PRO, int EPI, bool RAG>
__global__ __launch_bounds__(F32R_THREADS) void pw_f32p_kernel(const PwGemmArgs a) {
  static_assert(EPI == EPI_STATS || EPI == X3D_EPI_STORE || EPI == X3D_EPI_ADD || EPI == X3D_EPI_SWISH_BWD, "epilogue not built here (strided add, inference: pw_gemm_f32r.h)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef float T;
  constexpr int BM = MT * 32, BN = NT * 32, KC = F32R_KC, NW = F32R_THREADS / 64;
  constexpr int NTILE = MT * NT, TPW = (NTILE + NW - 1) / NW;
  constexpr bool HAS_SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  constexpr bool TWO = (PRO == PRO_BNBWD) || (PRO == PRO_TAIL);    // a second streamed tensor (a.x2)
  constexpr bool SIDE = (PRO == PRO_TAIL) || (PRO == PRO_AFFST);   // the activated input is also stored (a.ystore)
  constexpr bool GATE = (PRO == PRO_AFFINE) || (PRO == PRO_AFFST); // per-(sample, row) SE gate, loaded with the row
  constexpr int VPR = BN / 4;                                // 16-byte staging vectors per k row
  constexpr int RS = F32R_THREADS / VPR;                     // k rows per staging round
  constexpr int NXV = KC / RS;                               // vectors per thread and chunk
  constexpr int D = (NXV * (TWO ? 2 : 1) >= 8) ? 2 : 3;      // chunks in flight (register sets)
  static_assert(KC % RS == 0, "chunk must divide over the workgroup");
  const int WP = f32r_wpitch(a.K);
  const int nchunks = (a.K + KC - 1) / KC, Kp = nchunks * KC;
  float* Ws = smem;                                          // [BM][WP]
  float* Xs = smem + ((BM * WP + 3) & ~3);                   // [2][KC][BN]
  float* Pk = Xs + 2 * KC * BN;                              // [Kp][4]  prologue rows
  float* Em = Pk + Kp * 4;                                   // [BM][4]  epilogue rows

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  // workgroup -> (row group, tile range), XCD-aware: the hardware deals workgroups to the 8 XCDs round-robin, and the gy row groups
  // of one tile range read the SAME activations -- they take consecutive slots of ONE XCD (ids = xcd mod 8), so they run side by
  // side on its CUs and all but the first read the tiles from that XCD's L2 (as blockIdx.y they were gx workgroups apart: every
  // row group fetched the tensor again, 4-6 x its bytes)
  const int gy = (a.M + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rg = slot % gy, range = (slot / gy) * 8 + xcd;
  const int m0 = rg * BM;
  const int P = (int)a.P;
  const int tiles_per_n = (P + BN - 1) / BN;
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = range * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  F32P_STAMP(0);
  if (tile_begin >= tile_end) return;

  // ---- prologue rows (the same for every sample: the gate travels with the data)
  if constexpr (PRO != PRO_NONE) {
    for (int k = tid; k < Kp; k += F32R_THREADS) {
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
      if (k < a.K) {
        if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFST) {
          c0 = a.coef[k * 2]; c1 = a.coef[k * 2 + 1];
        } else if constexpr (PRO == PRO_TAIL) {   // s_c * x + (s_r | 1) * x2 + (t_c + t_r | 0), then ReLU
          c0 = a.coef[k * 2]; c1 = a.coef2 ? a.coef2[k * 2] : 1.0f;
          c2 = a.coef[k * 2 + 1] + (a.coef2 ? a.coef2[k * 2 + 1] : 0.f);
        } else {
          // (coef_fold: derived from the BatchNorm-backward sums; workgroup 0 publishes dgamma / dbeta / the table)
          bn_bwd_coef_load(a.coef, a.fold, k, blockIdx.x == 0, c0, c1, c2);
        }
      }
      Pk[k * 4] = c0; Pk[k * 4 + 1] = c1; Pk[k * 4 + 2] = c2;
    }
  }
  auto fill_em = [&](int n) __attribute__((always_inline)) {
    if constexpr (EPI == X3D_EPI_SWISH_BWD) {
      for (int m = tid; m < BM; m += F32R_THREADS) {
        const int gm = m0 + m;
        const bool ok = gm < a.M;
        Em[m * 4] = ok ? a.b_ss[gm * 2] : 0.f;
        Em[m * 4 + 1] = ok ? a.b_ss[gm * 2 + 1] : 0.f;
        Em[m * 4 + 2] = (ok && a.egate) ? a.egate[(long long)n * a.M + gm] : 1.0f;
      }
    }
  };

  // ---- buffer resources: whole tensors, 32-bit byte offsets (host: every tensor < F32P_OOB bytes)
  const int xbytes = a.N * a.K * P * 4, ybytes = a.N * a.M * P * 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((float*)a.x, 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((float*)(TWO ? a.x2 : a.x), 0, TWO ? xbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc((float*)(SIDE ? a.ystore : (void*)a.x), 0, SIDE ? xbytes : 0, 0x00020000);
  const bool has_gate = GATE && a.gate != nullptr;
  const __amdgpu_buffer_rsrc_t rgt = __builtin_amdgcn_make_buffer_rsrc((float*)(has_gate ? a.gate : (const float*)a.x), 0,
                                                                       has_gate ? a.N * a.K * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((float*)a.y, 0, ybytes, 0x00020000);
  constexpr bool EADD = (EPI == X3D_EPI_ADD) || (EPI == X3D_EPI_SWISH_BWD);
  const void* esrc_ = EPI == X3D_EPI_SWISH_BWD ? a.braw : a.add;
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((float*)((EADD && esrc_) ? esrc_ : a.y), 0,
                                                                      (EADD && esrc_) ? ybytes : 0, 0x00020000);

  // ---- staging role of the thread: rows kl0 + i * RS of a chunk, the 4 columns from pv * 4 of a tile
  const bool act_swish = a.act == X3D_ACT_SWISH;
  const float act_floor = a.act == X3D_ACT_RELU ? 0.f : -INFINITY;     // max(u, -inf) = u
  const int kl0 = tid / VPR, pv = tid % VPR;
  const int q4 = P & 3;                                      // RAG: elements of the last, partial vector of a row
  const int vrow = (kl0 * P + pv * 4) * 4;                   // byte offset of the thread's vector in (row 0, tile column 0)

  struct Cursor { int tile, kc, n, p0; };
  Cursor ld, cp;         // loads (D steps ahead); commit + MFMAs + epilogue
  ld.tile = tile_begin; ld.kc = 0; ld.n = tile_begin / tiles_per_n; ld.p0 = (tile_begin - ld.n * tiles_per_n) * BN;
  cp = ld;
  auto advance = [&](Cursor& c) __attribute__((always_inline)) {
    if (++c.kc == nchunks) {
      c.kc = 0; ++c.tile; c.p0 += BN;
      if (c.p0 >= P) { c.p0 = 0; ++c.n; }
    }
  };

  f32p_u32x4 xr[D][NXV], yr[D][TWO ? NXV : 1];
  float gr[D][GATE ? NXV : 1];
  // a thread's vector of the row's last, partial group of 4 columns (RAG) is loaded 4 - q4 elements early (inside the row) and
  // rotated at the commit: no load crosses the end of a row, none is conditional
  auto issue = [&](auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int pcol = ld.p0 + pv * 4;
    const bool colok = !(F32P_EXP & 8) && ld.tile < tile_end && pcol < P;
    // (the early start of a partial vector may reach in front of the tile: with row 0 / vector 0 the per-thread offset would go
    // negative -- 16 bytes move from the scalar to the per-thread offset; at p0 = 0 the partial vector is never vector 0, P >= 4)
    const int sb = (RAG && ld.p0 > 0) ? 16 : 0;
    int vo = vrow + sb;
    if constexpr (RAG) { if (pcol + 4 > P) vo -= (4 - q4) * 4; }
    const int klim = a.K - ld.kc * KC;                          // rows of this chunk inside K
    const int rowbase = ld.tile < tile_end ? ld.n * a.K + ld.kc * KC : 0;   // scalar (past the last tile: every offset is F32P_OOB)
#pragma unroll
    for (int i = 0; i < NXV; i++) {
      const int v = (colok && kl0 + i * RS < klim) ? vo : F32P_OOB;
      const int so = ((rowbase + i * RS) * P + ld.p0) * 4 - sb;
      xr[S][i] = __builtin_bit_cast(f32p_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v, so, 0));
      if constexpr (TWO) yr[S][i] = __builtin_bit_cast(f32p_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx2, v, so, 0));
      if constexpr (GATE) gr[S][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rgt, (kl0 + i * RS < klim) ? kl0 * 4 : F32P_OOB,
                                                                                                   (rowbase + i * RS) * 4, 0));
    }
  };
  // the commit of one step in pieces: pre = the step's scalars and prologue rows;
  // elem(i, e) = one element of staging vector i through the prologue; store(i) = vector i to LDS (and to ystore)
  struct CommitState { int pcol, klim, rowbase; bool colok, sh1, sh2, sh3; };
  CommitState cs;
  f32x4 ckv[PRO != PRO_NONE ? NXV : 1];
  float cv[NXV][4];
  auto commit_pre = [&]() __attribute__((always_inline)) {
    cs.pcol = cp.p0 + pv * 4;
    cs.colok = cp.tile < tile_end && cs.pcol < P;
    const bool part = RAG && cs.pcol + 4 > P;
    cs.sh1 = part && q4 == 1; cs.sh2 = part && q4 == 2; cs.sh3 = part && q4 == 3;
    cs.klim = a.K - cp.kc * KC;
    cs.rowbase = cp.n * a.K + cp.kc * KC;
    if constexpr (PRO != PRO_NONE) {
#pragma unroll
      for (int i = 0; i < NXV; i++) ckv[i] = *(const f32x4*)(Pk + (cp.kc * KC + kl0 + i * RS) * 4);     // zeros past K
    }
  };
  // element e of a loaded vector; the partial vector of a ragged row was loaded 4 - q4 elements early: rotate (selects, no branch)
  auto ragged = [&](const f32p_u32x4& v, int e) __attribute__((always_inline)) -> float {
    float x = __uint_as_float(v[e]);
    if constexpr (RAG) {
      if (e + 1 < 4) x = cs.sh3 ? __uint_as_float(v[(e + 1) & 3]) : x;
      if (e + 2 < 4) x = cs.sh2 ? __uint_as_float(v[(e + 2) & 3]) : x;
      if (e + 3 < 4) x = cs.sh1 ? __uint_as_float(v[(e + 3) & 3]) : x;
    }
    return x;
  };
  auto commit_elem = [&](auto SET, int i, int e) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    float x = ragged(xr[S][i], e);
    if constexpr (PRO == PRO_TAIL) {
      const float y = ragged(yr[S][TWO ? i : 0], e);
      x = fmaxf(ckv[i][0] * x + ckv[i][1] * y + ckv[i][2], 0.f);
    } else if constexpr (GATE) {
      // pw_prologue<PRO_AFFINE>'s arithmetic without its branch on the activation: the commit has to stay ONE basic block with
      // the MFMAs around it (both forms computed, the kernel-uniform one selected)
      const float g = has_gate ? gr[S][GATE ? i : 0] : 1.0f;
      const float u = (ckv[i][0] * x + ckv[i][1]) * g;
      const float sw = swishf_(u), mx = fmaxf(u, act_floor);
      x = act_swish ? sw : mx;
    } else if constexpr (PRO == PRO_BNBWD) {
      const float y = ragged(yr[S][TWO ? i : 0], e);
      x = ckv[i][0] * x + ckv[i][1] * y + ckv[i][2];
    }
    cv[i][e] = x;
  };
  auto commit_store = [&](int i, float* buf) __attribute__((always_inline)) {
    const int kl = kl0 + i * RS;
    if constexpr (SIDE) {     // y of the block below (the stem), kept for its other readers: written by the first row group
      const int so = ((cs.rowbase + i * RS) * P + cp.p0) * 4;
      const bool ok = rg == 0 && cs.colok && kl < cs.klim;
      if constexpr (RAG) {    // element stores: the row may end inside the vector
#pragma unroll
        for (int e = 0; e < 4; e++)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cv[i][e]), rst, (ok && cs.pcol + e < P) ? vrow + e * 4 : F32P_OOB, so, 0);
      } else {
        f32p_u32x4 sv;
#pragma unroll
        for (int e = 0; e < 4; e++) sv[e] = __float_as_uint(cv[i][e]);
        __builtin_amdgcn_raw_buffer_store_b128(sv, rst, ok ? vrow : F32P_OOB, so, 0);
      }
    }
    // (columns past P hold whatever the prologue makes of zeros: their output columns are never stored or summed; rows past K
    // are zeros times zero weights -- the prologue rows past K are zeros and relu / swish keep a zero)
    *(f32x4*)&buf[kl * BN + pv * 4] = (f32x4){cv[i][0], cv[i][1], cv[i][2], cv[i][3]};
  };
  constexpr int NPIECE = NXV * 4 + NXV;
  auto commit_piece = [&](auto SET, int p, float* buf) __attribute__((always_inline)) {
    if (p < NXV * 4) commit_elem(SET, p / 4, p % 4);
    else commit_store(p - NXV * 4, buf);
  };

  f32x16 acc[TPW];
  float st1[HAS_SUMS ? TPW : 1][16], st2[HAS_SUMS ? TPW : 1][16];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) {
      acc[s][j] = 0.f;
      if constexpr (HAS_SUMS) { st1[s][j] = 0.f; st2[s][j] = 0.f; }
    }
  // SWISH_BWD: per-(sample, row) sums, flushed when the workgroup moves to another sample
  auto flush_nc = [&](int n) __attribute__((always_inline)) {
    if constexpr (EPI == X3D_EPI_SWISH_BWD) {
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        const int id = wid + NW * s;
        if (NTILE % NW == 0 || id < NTILE) {
          const int mt = id / NT;
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const float s1 = half_wave_sum_hi(st1[s][j]);
            const float s2 = half_wave_sum_hi(st2[s][j]);
            const int m = m0 + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
            if (r == 16 && m < a.M) {
              double* d = a.nc_sums + ((long long)n * a.M + m) * 2;
              atomic_add_d(d, (double)s1);
              atomic_add_d(d + 1, (double)s2);
            }
            st1[s][j] = 0.f; st2[s][j] = 0.f;
          }
        }
      }
    }
  };

  // ---- epilogue of the tile at the compute cursor (pw_gemm.h's: the point index sits on the lane)
  // D[row][col]: col = lane & 31 (point), row = (j & 3) + 8 (j >> 2) + 4 (lane >> 5)
  // the epilogue's second operand (the Add input / the raw depthwise output of swish'), put in flight in FRONT of the MFMAs of a
  // tile's last chunk
  float eop[EADD ? TPW : 1][16];
  auto eload = [&]() __attribute__((always_inline)) {
    if constexpr (EADD) {
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        const int id = wid + NW * s;
        if (NTILE % NW == 0 || id < NTILE) {
          const int mt = id / NT, nt = id - mt * NT;
          const int p = cp.p0 + nt * 32 + r;
          const int mb = m0 + mt * 32 + 4 * half;
          const int vbase = p < P ? (mb * P + p) * 4 : F32P_OOB;
          const int mlim = a.M - mb;
          const int sbase = cp.n * a.M * P * 4;
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const int ro = (j & 3) + 8 * (j >> 2);
            eop[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(re, ro < mlim ? vbase : F32P_OOB, sbase + ro * P * 4, 0));
          }
        }
      }
    }
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    const int n = cp.n, p0 = cp.p0;
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + NW * s;
      if (NTILE % NW == 0 || id < NTILE) {
        const int mt = id / NT, nt = id - mt * NT;
        const int p = p0 + nt * 32 + r;
        const bool pok = p < P;
        const int mb = m0 + mt * 32 + 4 * half;                  // row of j = 0
        const int vbase = pok ? (mb * P + p) * 4 : F32P_OOB;
        const int mlim = a.M - mb;                               // rows (j & 3) + 8 (j >> 2) below it are real
        const int sbase = n * a.M * P * 4;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int ro = (j & 3) + 8 * (j >> 2);
          const int vo = ro < mlim ? vbase : F32P_OOB;
          const int so = sbase + ro * P * 4;
          float val = acc[s][j];
          if constexpr (EPI == EPI_STATS) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), ry, vo, so, 0);
            const float vm = pok ? val : 0.f;                    // (rows past M: zero weights, zero sums)
            st1[s][j] += vm;
            st2[s][j] += vm * vm;
          } else if constexpr (EPI == X3D_EPI_STORE) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), ry, vo, so, 0);
          } else if constexpr (EPI == X3D_EPI_ADD) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val + eop[EADD ? s : 0][j]), ry, vo, so, 0);
          } else if constexpr (EPI == X3D_EPI_SWISH_BWD) {
            const float b = eop[EADD ? s : 0][j];
            const float* em = Em + (mt * 32 + 4 * half + ro) * 4;
            const float u = em[0] * b + em[1];
            const float g = em[2];
            const float dv = val * swish_grad_(u * g);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), ry, vo, so, 0);
            const float dm = (pok && ro < mlim) ? dv : 0.f;
            st1[s][j] += dm;
            st2[s][j] += dm * b;
          }
          acc[s][j] = 0.f;
        }
      }
    }
  };

  // ---- the step loop.  Sub-step of step c: commit step c (register set c % D -> buffer c & 1) | barrier | issue step c + D into the
  // freed set | MFMAs of step c | epilogue if step c ends a tile
  int par = 0, n_pending = -1;
  auto sub = [&](auto SET) __attribute__((always_inline)) {
    float* buf = Xs + par * KC * BN;
    par ^= 1;
    if (!(F32P_EXP & 32)) {
      commit_pre();
#pragma unroll
      for (int p = 0; p < NPIECE; p++) commit_piece(SET, p, buf);
    }
    if (!(F32P_EXP & 64)) __syncthreads();                    // step c visible; every wave is past the MFMAs that read the other buffer
    issue(SET);
    advance(ld);
    const bool last = cp.kc == nchunks - 1 && cp.tile < tile_end;
    if (last) eload();
    const int k0 = cp.kc * KC;
    float av[TPW][KC / 2], bv[TPW][KC / 2];
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + NW * s;
      if (NTILE % NW == 0 || id < NTILE) {
        const int mt = id / NT, nt = id - mt * NT;
        const float* wp = Ws + (mt * 32 + r) * WP + k0 + half;
        const float* xp = buf + half * BN + nt * 32 + r;
#pragma unroll
        for (int i = 0; i < KC / 2; i++) { av[s][i] = wp[2 * i]; bv[s][i] = xp[2 * i * BN]; }
      }
    }
    // (tried: the commit of step c + 1 inside step c, its pieces dealt out between the 16 MFMAs of the dependent chain behind
    // scheduling fences -- 3 % SLOWER over the 14 layer shapes of tools/bench_f32r.py than this order, the same block left to the
    // compiler's order the same within 1 %: profiles/r06_f32p_sched.txt.  The matrix pipeline is saturated by the four waves of a
    // SIMD whenever they are in their MFMA runs; the commit does not hide inside them)
#pragma unroll
    for (int m = 0; m < TPW * KC / 2; m++) {
      const int s = m / (KC / 2), i = m % (KC / 2);
      if ((NTILE % NW == 0 || wid + NW * s < NTILE) && !((F32P_EXP & 1) && i != 0))
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][i], bv[s][i], acc[s], 0, 0, 0);
    }
    if (last && (!(F32P_EXP & 4) || acc[0][0] == 12345.f)) {
      epilogue();
      if constexpr (EPI == X3D_EPI_SWISH_BWD) {
        const int nn = cp.p0 + BN >= P ? cp.n + 1 : cp.n;
        n_pending = cp.n;
        if (nn != cp.n) {                // the next tile belongs to another sample: its sums and its gate row
          flush_nc(cp.n);
          n_pending = -1;
          if (cp.tile + 1 < tile_end) {
            __syncthreads();
            fill_em(nn);
            __syncthreads();
          }
        }
      }
    }
    advance(cp);
  };

  {
    issue(std::integral_constant<int, 0>());
    advance(ld);
    issue(std::integral_constant<int, 1>());
    advance(ld);
    if constexpr (D == 3) { issue(std::integral_constant<int, 2>()); advance(ld); }
  }
  // ---- the weight block, once: Ws[m][k] = w(k, m0 + m), zero padding (pw_gemm_f32r.h) -- behind the first D activation loads, whose
  // latency it covers (a 64 x 432 block is 110 KB: 6 us in front of the first barrier when it came first, in rounds of 4 loads)
  if (!(F32P_EXP & 2)) {
    const int total = BM * Kp;
    const bool vec4 = a.wsk == 1 ? ((a.K & 3) == 0 && (a.wsm & 3) == 0) : (a.wsm == 1 && (a.M & 3) == 0 && (a.wsk & 3) == 0);
    if (vec4 && (((uintptr_t)a.w) & 15) == 0) {
      constexpr int UW = 8;
      for (int base = 0; base < total / 4; base += F32R_THREADS * UW) {
        f32x4 wv[UW];
        int dk[UW], dm[UW];
#pragma unroll
        for (int u = 0; u < UW; u++) {
          const int i = (base + u * F32R_THREADS + tid) * 4;
          int k, m;
          if (a.wsk == 1) { m = i / Kp; k = i - m * Kp; }
          else { k = i / BM; m = i - k * BM; }
          const bool in = i < total;
          dk[u] = in ? k : -1; dm[u] = m;
          wv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (in && k < a.K && m0 + m < a.M) wv[u] = *(const f32x4*)(a.w + (long long)k * a.wsk + (long long)(m0 + m) * a.wsm);
        }
#pragma unroll
        for (int u = 0; u < UW; u++) {
          if (dk[u] < 0) continue;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (a.wsk == 1) Ws[dm[u] * WP + dk[u] + e] = wv[u][e];
            else Ws[(dm[u] + e) * WP + dk[u]] = wv[u][e];
          }
        }
      }
    } else {
      constexpr int UW = 8;
      for (int base = 0; base < total; base += F32R_THREADS * UW) {
        float wv[UW];
        int dst[UW];
#pragma unroll
        for (int u = 0; u < UW; u++) {
          const int i = base + u * F32R_THREADS + tid;
          int k, m;
          if (a.wsk == 1) { m = i / Kp; k = i - m * Kp; }
          else { k = i / BM; m = i - k * BM; }
          const bool in = i < total;
          dst[u] = in ? m * WP + k : -1;
          wv[u] = (in && k < a.K && m0 + m < a.M) ? a.w[(long long)k * a.wsk + (long long)(m0 + m) * a.wsm] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UW; u++) if (dst[u] >= 0) Ws[dst[u]] = wv[u];
      }
    }
  }
  fill_em(cp.n);
  F32P_STAMP(1);
  __syncthreads();                      // weights and tables in place
  F32P_STAMP(2);
  // (no early exit inside the unrolled body: the structurizer routes a break through the loop header, whose waits must then hold
  // for a set issued one sub-step ago -- vmcnt(0).  Up to D - 1 sub-steps past the last tile run on zeros: out-of-range loads,
  // no epilogue)
  const int nsteps = (tile_end - tile_begin) * nchunks;
  F32P_STAMP(3);
  for (int step = 0; step < nsteps; step += D) {
    F32P_STAMP(4 + step);
    sub(std::integral_constant<int, 0>());
    sub(std::integral_constant<int, 1>());
    if constexpr (D == 3) sub(std::integral_constant<int, 2>());
  }

  F32P_STAMP(60);
  if constexpr (EPI == X3D_EPI_SWISH_BWD) { if (n_pending >= 0) flush_nc(n_pending); }
  if constexpr (EPI == EPI_STATS) {
    // the NT waves of a row block hold partial sums of the same channels: added up in LDS (in double, as the atomics would),
    // then one pair of atomics per channel and workgroup
    float* red = Xs;                    // [NT][BM][2], free once every wave is past its last MFMA
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + NW * s;
      if (NTILE % NW == 0 || id < NTILE) {
        const int mt = id / NT, nt = id - mt * NT;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const float s1 = half_wave_sum_hi(st1[s][j]);
          const float s2 = half_wave_sum_hi(st2[s][j]);
          const int ml = mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
          if (r == 16) { red[(nt * BM + ml) * 2] = s1; red[(nt * BM + ml) * 2 + 1] = s2; }
        }
      }
    }
    __syncthreads();
    if (tid < BM * 2 && a.stats && !(F32P_EXP & 16)) {
      const int ml = tid >> 1, q = tid & 1, m = m0 + ml;
      if (m < a.M) {
        double t = 0.0;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) t += (double)red[(nt * BM + ml) * 2 + q];
        atomic_add_d(&stats_replica(a.stats, a.M, range)[m * 2 + q], t);
      }
    }
  }
  F32P_STAMP(61);
}

// X3D_PW_F32P=0: A/B hook (pw_gemm_f32r.h's kernel)
static inline bool f32p_enabled() { return x3d_env_int("X3D_PW_F32P", 1) != 0; }

template <int MT, int NT, int PRO, int EPI, bool RAG>
static int f32p_launch_cfg(PwGemmArgs& a, hipStream_t st) {
  constexpr int BM = MT * 32, BN = NT * 32;
  const size_t lds = f32r_lds_bytes(a.K, MT, NT);
  X3D_DESCRIBE("pw_f32p_kernel<%d, %d, %d, %d, %d>", MT, NT, PRO, EPI, (int)RAG);
  auto kern = pw_f32p_kernel<MT, NT, PRO, EPI, RAG>;
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
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, F32R_THREADS, lds) != hipSuccess || nb < 1) nb = 1;
    slots = nb * x3d_device_cus();
    if (occ_n < 8) { occ_lds[occ_n] = lds; occ_slots[occ_n] = slots; occ_n++; }
  }
  const int gy = ceil_div(a.M, BM);
  const long long total_tiles = ceil_div_ll(a.P, BN) * a.N;
  long long per_group = slots / gy;                    // one round of persistent workgroups over all row groups
  if (per_group < 1) per_group = 1;
  long long tpb = ceil_div_ll(total_tiles, per_group);
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = (int)tpb;
  const long long gx = ceil_div_ll(total_tiles, tpb);
  const long long grid = ceil_div_ll(gx, 8) * 8 * gy;  // (tile ranges padded to the 8 XCDs: a workgroup past the last range returns at once)
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(F32R_THREADS), lds, st, a);
  X3D_LAUNCH_CHECK("pw_f32p");
  return X3D_OK;
}

// returns -1 when the launch is not covered (the caller goes on to pw_gemm_f32r.h / pw_gemm.h)
template <int PRO, int EPI>
static int f32p_try(PwGemmArgs& a, hipStream_t st) {
  if (!f32p_enabled() || !f32r_enabled() || a.stride != 1 || a.P < 4 || a.Pin != a.P) return -1;
  const long long big = ((long long)a.N * (a.K > a.M ? a.K : a.M) + 256) * a.P * 4;
  if (big >= (long long)F32P_OOB) return -1;                    // 32-bit buffer offsets
  if ((((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.x2 | (uintptr_t)a.ystore | (uintptr_t)a.add | (uintptr_t)a.braw) & 3) != 0) return -1;
  int MT = 0, NT = 0;
  constexpr bool SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  if (!f32r_shape(a, SUMS ? 2 : 4, &MT, &NT)) return -1;
  const bool rag = (a.P & 3) != 0;      // rows end inside a 16-byte vector (16-byte accesses at 4-byte addresses are fine: unaligned access mode)
#define F32P_CASE(M_, N_) if (MT == M_ && NT == N_) return rag ? f32p_launch_cfg<M_, N_, PRO, EPI, true>(a, st) : f32p_launch_cfg<M_, N_, PRO, EPI, false>(a, st);
  F32P_CASE(1, 8) F32P_CASE(2, 4)
  if constexpr (!SUMS) { F32P_CASE(3, 4) F32P_CASE(4, 4) }
#undef F32P_CASE
  return -1;
}
