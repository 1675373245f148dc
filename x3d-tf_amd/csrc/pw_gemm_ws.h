// Weights-streamed small-tile variant of the bf16 pointwise GEMM for the deep, narrow layers (X3D stage 5:
// 192 <-> 432 channels on 784 points per sample).
//
// The resident-panel kernel (pw_gemm_bf16.h) keeps a 32-row weight panel in LDS and tiles 128 points: with K = 432
// that is six row blocks per point tile (the prologue -- swish or BN-backward per element -- is repeated six times),
// only 392 point tiles per launch, and 7 K-chunks of serialised stage -> barrier -> MFMA per tile.  Here a workgroup
// owns a tile of 32 points with ALL K rows resident in LDS (K * 64 bytes: the prologue runs once per element), and
// walks every 32-row block of the output: the A operand (weights) is read straight from the packed bf16 panel in
// global memory -- 16 bytes per lane, L2 resident (<= 166 KB per layer) -- in double-buffered groups of k-steps, the
// B operand with ds_read_b64_tr_b16 (pitch 64 B = exactly one bank segment per row).  1,600 tiles per launch instead
// of 392, no redundant prologue, no per-workgroup panel copy.
//   wave w computes row blocks w, w+4, ...; each 32x32 result goes through a wave-private LDS slab so the epilogue
//   (statistics / residual add / swish' with per-(n,c) sums) works on 16-point row pieces with 16-byte accesses.
#pragma once
#include <type_traits>

#include "pw_gemm_bf16.h"

#define WS_BN 32
#define WS_G 8                 // k-steps per A-operand group (double buffered: 2 * 8 * 4 VGPRs)
#define WS_OP 36               // slab pitch (floats)
#define WS_MAXMT 4             // row blocks per wave (M <= 512)
#ifndef WS_OCC
#define WS_OCC 2
#endif

template <typename H, int PRO, int EPI>
__global__ __launch_bounds__(256, WS_OCC) void pw_gemm_ws_kernel(const PwGemmArgs a) {
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4; typedef typename HV<H>::x2 hx2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BN = WS_BN, G = WS_G, OP = WS_OP;
  constexpr bool HAS_SUMS = (EPI == EPI_STATS) || (EPI == X3D_EPI_SWISH_BWD);
  constexpr int CSW = (PRO == PRO_AFFINE) ? 2 : 4;
  constexpr int NSV = 7;                                // staging vectors per thread: Kp * 4 / 256 <= 7 (Kp <= 448)
  const int Kp = a.KC, WP = Kp + 8;
  const int ksteps = Kp / 16;
  H* Xs = (H*)smem_raw;                                                // [Kp][32]
  float* Cs = (float*)(smem_raw + (size_t)Kp * 64);                          // [Kp][CSW]
  float* Os = (float*)(smem_raw + (size_t)Kp * 64 + (size_t)Kp * 16);        // [4 waves][32][OP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int mt = (a.M + 31) >> 5;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  float* myOs = Os + wid * 32 * OP;

  auto fill_coef = [&](int n) {
    if constexpr (PRO != PRO_NONE) {
      for (int k = tid; k < Kp; k += 256) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        if (k < a.K) {
          if constexpr (PRO == PRO_AFFINE) {
            const float g = a.gate ? a.gate[(long long)n * a.K + k] : 1.0f;
            c[0] = a.coef[k * 2] * g;
            c[1] = a.coef[k * 2 + 1] * g;
          } else {
            c[0] = a.coef[k * 4]; c[1] = a.coef[k * 4 + 1]; c[2] = a.coef[k * 4 + 2];
          }
        }
        if constexpr (CSW == 2) *(float2*)&Cs[k * 2] = make_float2(c[0], c[1]);
        else *(f32x4*)&Cs[k * 4] = c;
      }
    }
  };
  if (tile_begin < tile_end) fill_coef(tile_begin / tiles_per_n);

  // ---- staging: vector v = tid + 256*i -> row v >> 2, 8 points at unit v & 3
  hx8 xr[NSV], yr[PRO == PRO_BNBWD ? NSV : 1];
  auto issue_loads = [&](int tile) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * 256;
      const int k = v >> 2;
      const long long p = p0 + (v & 3) * 8;
      hx8 z;
#pragma unroll
      for (int e = 0; e < 8; e++) z[e] = (H)0.f;
      xr[i] = z;
      if constexpr (PRO == PRO_BNBWD) yr[i] = z;
      if (k < a.K && p < a.P) {
        const long long o = ((long long)n * a.K + k) * a.Pin + p;
        xr[i] = *(const hx8*)((const T*)a.x + o);
        if constexpr (PRO == PRO_BNBWD) yr[i] = *(const hx8*)((const T*)a.x2 + o);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * 256;
      const int k = v >> 2;
      if (k >= Kp) continue;
      H* dst = &Xs[k * BN + (v & 3) * 8];
      if constexpr (PRO == PRO_NONE) {
        *(hx8*)dst = xr[i];
      } else {
        float val[8];
#pragma unroll
        for (int e = 0; e < 8; e++) val[e] = (float)xr[i][e];
        if constexpr (PRO == PRO_AFFINE) {
          const float2 cf = *(const float2*)&Cs[k * 2];
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] = cf.x * val[e] + cf.y;
          act_vec<8>(val, a.act);
        } else {
          const f32x4 cf = *(const f32x4*)&Cs[k * 4];
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] = cf[0] * val[e] + cf[1] * (float)yr[i][e] + cf[2];
        }
        VecIO<H, 8>::store(dst, val);
      }
    }
  };

  // per-lane partial sums: lane owns row (lane >> 1) of each of this wave's row blocks, 16 points
  float st1[HAS_SUMS ? WS_MAXMT : 1], st2[HAS_SUMS ? WS_MAXMT : 1];
  if constexpr (HAS_SUMS) {
#pragma unroll
    for (int i = 0; i < WS_MAXMT; i++) { st1[i] = 0.f; st2[i] = 0.f; }
  }
  auto flush_sums = [&](int n) {
    if constexpr (HAS_SUMS) {
#pragma unroll
      for (int i = 0; i < WS_MAXMT; i++) {
        const int mi = wid + 4 * i;
        // the two lanes of a row: quad_perm [1,0,3,2]
        const float s1 = st1[i] + dpp_get<0xB1, 0xF>(st1[i]), s2 = st2[i] + dpp_get<0xB1, 0xF>(st2[i]);
        const int m = mi * 32 + (lane >> 1);
        if (mi < mt && (lane & 1) == 0 && m < a.M) {
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

  // transposed-read lane geometry inside the 32-point tile
  const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_row = 8 * (g16 >> 1) + q;
  const int tr_col = 16 * (g16 & 1) + 4 * pp;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

  if (tile_begin < tile_end) issue_loads(tile_begin);
  int n_prev = tile_begin < tile_end ? tile_begin / tiles_per_n : 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    if (n != n_prev) {
      if constexpr (EPI == X3D_EPI_SWISH_BWD) flush_sums(n_prev);
      if constexpr (PRO == PRO_AFFINE) { if (a.gate) fill_coef(n); }
    }
    n_prev = n;
    __syncthreads();            // every wave is done with the previous tile's Xs (and sees the coefficient table)
    commit();
    __syncthreads();
    if (tile + 1 < tile_end) issue_loads(tile + 1);

    for (int i = 0; i < WS_MAXMT; i++) {
      const int mi = wid + 4 * i;
      if (mi >= mt) break;
      // ---- 32x32 block mi: A from the packed panel in global memory, groups of G k-steps, double buffered
      // tiled image behind the row-major one (pw_pack.hip): k-step ks of row block mi = 64 lanes x 16 B, contiguous
      const H* wrow = (const H*)a.wp + (long long)a.wp_rows * WP + ((long long)mi * ksteps * 64 + lane) * 8;
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 16; j++) acc[j] = 0.f;
      hx8 A0[G], A1[G];
      // a group of G k-steps is either whole (no per-step test, unconditional loads: every group but possibly the last)
      // or ragged; one uniform branch per GROUP picks the variant -- the per-step tests were 2 scalar branches per MFMA
      auto loadA = [&](hx8 (&A)[G], int ks0, auto FULL) {
#pragma unroll
        for (int j = 0; j < G; j++) {
          if constexpr (decltype(FULL)::value) {
            A[j] = *(const hx8*)(wrow + (ks0 + j) * 512);
          } else {
            hx8 z;
#pragma unroll
            for (int e = 0; e < 8; e++) z[e] = (H)0.f;
            A[j] = (ks0 + j < ksteps) ? *(const hx8*)(wrow + (ks0 + j) * 512) : z;
          }
        }
      };
      auto mmaA = [&](const hx8 (&A)[G], int ks0, auto FULL) {
#pragma unroll
        for (int j = 0; j < G; j++) {
          if (decltype(FULL)::value || ks0 + j < ksteps) {
            const int kk = (ks0 + j) * 16;
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[(kk + tr_row) * BN + tr_col]));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[(kk + tr_row + 4) * BN + tr_col]));
            const s16x8 bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            acc = mfma16<H>(A[j], __builtin_bit_cast(hx8, bs), acc);
          }
        }
      };
      auto load_g = [&](hx8 (&A)[G], int ks0) {
        if (ks0 + G <= ksteps) loadA(A, ks0, std::true_type{});
        else if (ks0 < ksteps) loadA(A, ks0, std::false_type{});
      };
      auto mma_g = [&](const hx8 (&A)[G], int ks0) {
        if (ks0 + G <= ksteps) mmaA(A, ks0, std::true_type{});
        else if (ks0 < ksteps) mmaA(A, ks0, std::false_type{});
      };
      load_g(A0, 0);
      for (int ks0 = 0; ks0 < ksteps; ks0 += 2 * G) {
        load_g(A1, ks0 + G);
        mma_g(A0, ks0);
        load_g(A0, ks0 + 2 * G);
        mma_g(A1, ks0 + G);
      }

      // ---- epilogue through the wave-private slab: lane -> row lane >> 1, points 16*(lane & 1) .. +15
#pragma unroll
      for (int j = 0; j < 16; j++) myOs[((j & 3) + 8 * (j >> 2) + 4 * half) * OP + r] = acc[j];
      const int row = lane >> 1, c0 = 16 * (lane & 1);
      const int m = mi * 32 + row;
      if (m < a.M) {
        float sb = 0.f, tb = 0.f, gt = 1.f;
        if constexpr (EPI == X3D_EPI_SWISH_BWD) {
          sb = a.b_ss[m * 2]; tb = a.b_ss[m * 2 + 1];
          gt = a.egate ? a.egate[(long long)n * a.M + m] : 1.0f;
        }
        if constexpr (EPI == EPI_BNADD) bnadd_coef(a, m, true, sb, tb, gt);
#pragma unroll
        for (int hv = 0; hv < 2; hv++) {
          const long long p = p0 + c0 + 8 * hv;
          if (p >= a.P) continue;                       // P % 8 == 0: a vector of 8 points is inside or outside
          const long long o = ((long long)n * a.M + m) * a.P + p;
          float val[8];
          {
            const f32x4 v0 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv], v1 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv + 4];
#pragma unroll
            for (int e = 0; e < 4; e++) { val[e] = v0[e]; val[4 + e] = v1[e]; }
          }
          if constexpr (EPI == X3D_EPI_ADD) {
            float ad[8];
            VecIO<T, 8>::load((const T*)a.add + o, ad);
#pragma unroll
            for (int e = 0; e < 8; e++) val[e] += ad[e];
          } else if constexpr (EPI == EPI_BNADD) {
            float ad[8];
#pragma unroll
            for (int e = 0; e < 8; e++) ad[e] = 0.f;
            if (a.add) VecIO<T, 8>::load((const T*)a.add + o, ad);
            const float lo = a.eact == X3D_ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
            for (int e = 0; e < 8; e++) val[e] = fmaxf(sb * val[e] + tb + gt * ad[e], lo);
          } else if constexpr (EPI == X3D_EPI_SWISH_BWD) {
            float b[8];
            VecIO<T, 8>::load((const T*)a.braw + o, b);
            const SwishCoef sc_ = swish_coef(sb, tb, gt);
#pragma unroll
            for (int e = 0; e < 8; e++) {
              float xh_, d_;
              swish_bwd_(sc_, b[e], xh_, d_);
              const float dv = val[e] * d_;
              val[e] = dv;
              st1[i] += dv;
              st2[i] += dv * b[e];
            }
          }
          if constexpr (EPI == EPI_STATS) {
#pragma unroll
            for (int e = 0; e < 8; e++) { st1[i] += val[e]; st2[i] += val[e] * val[e]; }
          }
          VecIO<T, 8>::store((T*)a.y + o, val);
        }
      }
    }
  }
  if (tile_begin < tile_end) flush_sums(n_prev);
}

static inline size_t pw_ws_lds_bytes(int K) {
  const int Kp = (K + 15) & ~15;
  return (size_t)Kp * 64 + (size_t)Kp * 16 + (size_t)4 * 32 * WS_OP * 4;
}

// the deep, narrow layers: few 128-point tiles per launch and a wide contraction or output
static inline bool pw_ws_applies(const PwGemmArgs& a, int vec, int ovec) {
  const int e_ws = x3d_env_int("X3D_PW_WS", -1);   // A/B switch: 0 = never, 1 = whenever legal
  if (e_ws == 0) return false;
  if (!a.wp || vec < 8 || ovec < 8 || a.stride != 1 || (a.P % 8) != 0) return false;
  if (a.K > 448 || a.M > 128 * WS_MAXMT) return false;
  if (e_ws == 1) return true;
  // measured on X3D-M stage 5 (r01g): wins where the resident-panel kernel needs 32-row panels and repeats the prologue
  // per row block (K = 432 -> M = 192: 94 -> 71 us forward, 84 -> 67 us dgrad); loses for narrow K / wide M (K = 192 ->
  // M = 432: 37 -> 60 us), where streaming all of W per 32-point tile (tiles x |W| = 265 MB through L2) is the bound
  const long long tiles128 = ceil_div_ll(a.P, 128) * a.N;
  return tiles128 <= 1024 && a.K >= 320 && a.M <= 256;
}

template <typename H, int PRO, int EPI>
static int pw_ws_launch(PwGemmArgs& a, hipStream_t st) {
  a.KC = (a.K + 15) & ~15;
  const size_t lds = pw_ws_lds_bytes(a.K);
  X3D_DESCRIBE("pw_gemm_ws_kernel<%s, %d, %d>", HV<H>::name, PRO, EPI);
  auto kern = pw_gemm_ws_kernel<H, PRO, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
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
  const long long total_tiles = ceil_div_ll(a.P, WS_BN) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_gemm_ws: too many tiles");
  long long tpb = ceil_div_ll(total_tiles, slots);
  if (tpb < 2) tpb = 2;
  a.tiles_per_block = (int)tpb;
  const long long gx = ceil_div_ll(total_tiles, tpb);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_gemm_ws");
  return X3D_OK;
}
