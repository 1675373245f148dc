// Weights-stationary FUSED backward of the stage-4 `a` conv (96 block channels -> 216 inner channels on 14 x 14 planes;
// reference model.py:246-253 through SURVEY appendix A): data gradient with the residual add (and, optionally, the
// Add + ReLU backward of the block below) AND the weight gradient, one pass over g / a_raw / x.
//
// Until round 3 this layer ran as x3d_pw_dgrad + x3d_pw_wgrad (56 + 56 us, ten launches each per X3D-M step, every one
// streaming g and a_raw -- 2 x 216 rows -- to rebuild dY) followed by x3d_tail_bwd (27 us) on the result: the sliced fused
// kernel only takes dY tiles of <= 128 rows.  Same organisation as pw_bwd_wst.hip, roles swapped:
//   * dX = W^T dY: M = Ci <= 96 (NW = 3 MFMA waves, one 32-row block each), K = Co = 216 (KS = 14 k-steps; 56 VGPRs of
//     stationary weights per wave); the dY tile [Co][32 points] is staged once (BN_a backward prologue, fp32) in two LDS
//     layouts -- [k][32] for the transposed B operand of dX, [co][32 + 8] for the row-wise A operand of dW;
//   * the x tile [Ci][32 + 8] (the conv input = the output y of the block below) is the B operand of dW and the ReLU mask of
//     the folded tail;
//   * dW[co][ci] += dY[co][:] . x[ci][:]: CT = 7 row tiles of Co x 3 of Ci = 21 tiles, wave w (of 7) owns cot = w: 3 x 16
//     accumulator VGPRs across all tiles of the workgroup, one fp32 atomic flush;
//   * epilogue (waves 0..2, private slabs): dx = dX + add; with tail_c: dx *= [x > 0] and the BN_c / BN_r backward sums of the
//     block below (exactly what x3d_tail_bwd computes from the stored dx).
// One barrier per tile; every global access of the tile loop is unconditional (clamped loads, bounds-checked buffer stores).
#include <stdlib.h>

#include "pw_gemm_ws.h"

struct PwBwdWstaArgs {
  const void* g; const void* yraw; const float* coef;   // dY = A*g + B*yraw + C   rows = Co
  const void* wp; int wp_rows;                          // dgrad panel (tiled image behind the row-major one)
  void* dx;                                             // [N][Ci][P]
  const void* add;                                      // [N][Ci][P]
  const void* x;                                        // [N][Ci][P] conv input
  float* dw;                                            // [Co][Ci]
  const void* tail_c; const void* tail_r; double* tail_sums_c; double* tail_sums_r;
  int N, Co, Ci;
  long long P;
  int tiles_per_block;
  BnBwdFold fold;                                       // sums != NULL: the coefficient table is derived from the BatchNorm-backward sums here (x3d_hip.h coef_fold)
  float* slab;                                          // NULL | per-workgroup partial dW slabs [gridDim.x][Co][Ci] (plain stores)
  int hot;                                              // experiments build only (X3D_PW_BWD_HOT=1): every tile load re-reads the FIRST tile
  int noflush;                                          // experiments build only (X3D_PW_BWD_NOFLUSH=1): timing without the dW flush
};

#define BWA_RP 40    // pitch (elements) of the row-read tiles: 80 B = 5 units, odd -> b128 rows conflict-free

// TAIL: 0 = plain residual add, 1 = + folded tail backward (identity shortcut below), 2 = ... with a shortcut conv below
template <typename H, int NW, int KS, int CT, int TAIL>
__global__ __launch_bounds__(512, 2) void pw_bwd_wsta_kernel(const PwBwdWstaArgs a) {
  typedef typename HV<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BN = 32, OP = WS_OP, NT = 512, Kp = KS * 16, WP = Kp + 8, CoP = CT * 32, CiP = NW * 32, RP = BWA_RP;
  static_assert(CoP >= Kp && CT <= 8, "dY rows / weight-gradient waves");
  constexpr int NSV = (Kp * 4 + NT - 1) / NT;            // dY staging vectors (8 points) per thread and tensor
  constexpr size_t YT_B = (size_t)2 * Kp * 64, YR_B = (size_t)2 * CoP * RP * 2, XR_B = (size_t)2 * CiP * RP * 2;
  H* Yt = (H*)smem_raw;                                  // [2][Kp][32]
  H* Yr = (H*)(smem_raw + YT_B);                         // [2][CoP][RP]
  H* Xr = (H*)(smem_raw + YT_B + YR_B);                  // [2][CiP][RP]
  float* Cs = (float*)(smem_raw + YT_B + YR_B + XR_B);   // [Kp][4]
  float* Os = Cs + Kp * 4;                               // [NW][32][OP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  if (tile_begin >= tile_end) return;
  float* myOs = Os + (wid < NW ? wid : 0) * 32 * OP;
  const bool xw = wid < NW && wid * 32 < a.Ci;           // dX / epilogue wave
  const bool ww = wid < CT && wid * 32 < a.Co;           // weight-gradient wave (cot = wid)

  {   // one-time set-up: zero both row-read tiles (padding rows stay zero), the BN-backward coefficient table
    hx8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (H)0.f;
    for (int i = tid; i < (int)((YR_B + XR_B) / 16); i += NT) ((hx8*)Yr)[i] = z;
    for (int k = tid; k < Kp; k += NT) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      if (k < a.Co) {
        float cA_, cB_, cC_;
        bn_bwd_coef_load(a.coef, a.fold, k, blockIdx.x == 0 && blockIdx.y == 0, cA_, cB_, cC_);
        c[0] = cA_; c[1] = cB_; c[2] = cC_;
      }
      *(f32x4*)&Cs[k * 4] = c;
    }
  }

  // ---- the stationary operand: W^T rows (input channels) 32 * wid .., all Kp output channels
  hx8 A[KS];
  if (xw) {
    const H* wt = (const H*)a.wp + (long long)a.wp_rows * WP + ((long long)wid * KS * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) A[ks] = *(const hx8*)(wt + ks * 512);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int e = 0; e < 8; e++) A[ks][e] = (H)0.f;
  }

  // ---- staging: dY vectors v = tid + NT * i -> row v >> 2, unit v & 3; x vectors v = tid -> row tid >> 2 (tid < 4 * CiP)
  hx8 g0[NSV], y0[NSV], g1[NSV], y1[NSV], x0, x1;
  auto issue_loads = [&](int tile_, hx8 (&gr)[NSV], hx8 (&yr)[NSV], hx8& xr) __attribute__((always_inline)) {
    const int tile = a.hot ? tile_begin : min(tile_, tile_end - 1);
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * NT;
      const int k = v >> 2;
      const long long p = p0 + (v & 3) * 8;
      const long long o = (k < a.Co && p < a.P) ? ((long long)n * a.Co + k) * a.P + p : 0;
      gr[i] = *(const hx8*)((const T*)a.g + o);
      yr[i] = *(const hx8*)((const T*)a.yraw + o);
    }
    {
      const int mrow = tid >> 2;
      const long long p = p0 + (tid & 3) * 8;
      const long long o = (mrow < a.Ci && p < a.P) ? ((long long)n * a.Ci + mrow) * a.P + p : 0;
      xr = *(const hx8*)((const T*)a.x + o);
    }
  };
  auto commit = [&](int tile, int buf, const hx8 (&gr)[NSV], const hx8 (&yr)[NSV], const hx8& xr) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * NT;
      const int k = v >> 2;
      if (k >= Kp) continue;
      const bool ok = k < a.Co && p0 + (v & 3) * 8 < a.P;
      const f32x4 cf = *(const f32x4*)&Cs[k * 4];
      hx8 hv;
#pragma unroll
      for (int e = 0; e < 8; e++) hv[e] = (H)(ok ? cf[0] * (float)gr[i][e] + cf[1] * (float)yr[i][e] + cf[2] : 0.f);   // (C must not leak into the padding)
      *(hx8*)&Yt[(buf * Kp + k) * BN + (v & 3) * 8] = hv;
      *(hx8*)&Yr[(buf * CoP + k) * RP + (v & 3) * 8] = hv;
    }
    const int mrow = tid >> 2;
    if (mrow < CiP) {
      const bool ok = mrow < a.Ci && p0 + (tid & 3) * 8 < a.P;
      hx8 hv = xr;
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; e++) hv[e] = (H)0.f;
      }
      *(hx8*)&Xr[(buf * CiP + mrow) * RP + (tid & 3) * 8] = hv;
    }
  };

  // ---- epilogue operands (waves 0..NW-1): lane -> row lane >> 1 of this wave's block, points 16 * (lane & 1) .. + 15
  const int row = lane >> 1, c0 = 16 * (lane & 1);
  const int m = wid * 32 + row;                          // this lane's input channel
  const bool mrow_ok = xw && m < a.Ci;
  constexpr int NEO = 1 + (TAIL >= 1) + (TAIL == 2);     // add, tail_c, tail_r
  hx8 eo0[NEO][2], eo1[NEO][2];
  auto issue_epi = [&](int tile_, hx8 (&eo)[NEO][2]) __attribute__((always_inline)) {
    const int tile = a.hot ? tile_begin : min(tile_, tile_end - 1);
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int hv = 0; hv < 2; hv++) {
      const long long p = p0 + c0 + 8 * hv;
      const long long o = (mrow_ok && p < a.P) ? ((long long)n * a.Ci + m) * a.P + p : 0;
      eo[0][hv] = *(const hx8*)((const T*)a.add + o);
      if constexpr (TAIL >= 1) eo[1][hv] = *(const hx8*)((const T*)a.tail_c + o);
      if constexpr (TAIL == 2) eo[2][hv] = *(const hx8*)((const T*)a.tail_r + o);
    }
  };
  float tg = 0.f, tgc = 0.f, tgr = 0.f;                  // TAIL: per-channel sums of this lane's row

  f32x16 acc_dw[NW];
#pragma unroll
  for (int s = 0; s < NW; s++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc_dw[s][e] = 0.f;

  const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_off = (8 * (g16 >> 1) + q) * BN + 16 * (g16 & 1) + 4 * pp;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;

  issue_loads(tile_begin, g0, y0, x0);
  issue_epi(tile_begin, eo0);
  __syncthreads();                                       // coefficient table, zeroed padding
  commit(tile_begin, 0, g0, y0, x0);
  __syncthreads();
  issue_loads(tile_begin + 1, g1, y1, x1);
  issue_loads(tile_begin + 2, g0, y0, x0);

  auto step = [&](int tile, int cur, hx8 (&gr)[NSV], hx8 (&yr)[NSV], hx8& xr, hx8 (&eo)[NEO][2], hx8 (&eon)[NEO][2])
      __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    issue_epi(tile + 1, eon);                            // a whole step to land
    __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc((T*)a.dx + (long long)n * a.Ci * a.P, 0,
                                                                   (int)((long long)a.Ci * a.P * 2), 0x00020000);
    // ---- dX = W^T dY
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    if (xw) {
      const H* xb = Yt + cur * (Kp * BN) + tr_off;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb + ks * 16 * BN));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb + (ks * 16 + 4) * BN));
        const s16x8 bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc = mfma16<H>(A[ks], __builtin_bit_cast(hx8, bs), acc);
      }
    }
    // ---- dW tiles (cot = wid, cit = 0..NW-1): A = this wave's dY rows, B = the x rows (both row-read tiles of `cur`)
    if (ww) {
      const H* yrow = Yr + (cur * CoP + wid * 32 + r) * RP + 8 * half;
      const H* xrow = Xr + (cur * CiP + r) * RP + 8 * half;
#pragma unroll
      for (int ks = 0; ks < BN / 16; ks++) {
        const hx8 af = *(const hx8*)(yrow + ks * 16);
#pragma unroll
        for (int s = 0; s < NW; s++) {
          const hx8 bf = *(const hx8*)(xrow + s * 32 * RP + ks * 16);
          acc_dw[s] = mfma16<H>(af, bf, acc_dw[s]);
        }
      }
    }
    // ---- the next tile goes into the other buffers (before the epilogue: its wait covers loads only)
    if (tile + 1 < tile_end) commit(tile + 1, cur ^ 1, gr, yr, xr);

    if (xw) {
      // ---- epilogue through the wave-private slab
#pragma unroll
      for (int e = 0; e < 16; e++) myOs[((e & 3) + 8 * (e >> 2) + 4 * half) * OP + r] = acc[e];
#pragma unroll
      for (int hv = 0; hv < 2; hv++) {
        const long long p = p0 + c0 + 8 * hv;
        const bool ok = mrow_ok && p < a.P;              // P % 8 == 0: a vector of 8 points is inside or outside
        float val[8];
        {
          const f32x4 v0 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv], v1 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv + 4];
#pragma unroll
          for (int e = 0; e < 4; e++) { val[e] = v0[e]; val[4 + e] = v1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) val[e] += (float)eo[0][hv][e];
        if constexpr (TAIL >= 1) {   // Add + ReLU backward of the block this gradient leaves (its output is x)
          const hx8 xv = *(const hx8*)&Xr[(cur * CiP + m) * RP + c0 + 8 * hv];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float gm = (ok && (float)xv[e] > 0.f) ? round_to<T>(val[e]) : 0.f;   // the sums describe dx as stored
            val[e] = gm;
            tg += gm;
            tgc += gm * (float)eo[1][hv][e];
            if constexpr (TAIL == 2) tgr += gm * (float)eo[2][hv][e];
          }
        }
        hx8 ov;
#pragma unroll
        for (int e = 0; e < 8; e++) ov[e] = (H)val[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, ov), dxr,
                                               ok ? (unsigned)(((long long)m * a.P + p) * 2) : 0x80000000u, 0, 0);
      }
    }
    __syncthreads();                                     // one barrier per tile
    issue_loads(tile + 3, gr, yr, xr);
  };
  for (int tile = tile_begin; tile < tile_end; tile += 2) {
    step(tile, 0, g1, y1, x1, eo0, eo1);
    if (tile + 1 < tile_end) step(tile + 1, 1, g0, y0, x0, eo1, eo0);
  }

  if constexpr (TAIL >= 1) {
    const float s0 = tg + dpp_get<0xB1, 0xF>(tg), s1 = tgc + dpp_get<0xB1, 0xF>(tgc);   // the two lanes of a row
    float s2 = 0.f;
    if constexpr (TAIL == 2) s2 = tgr + dpp_get<0xB1, 0xF>(tgr);
    if (mrow_ok && (lane & 1) == 0) {
      atomic_add_d(&a.tail_sums_c[m * 2], (double)s0);
      atomic_add_d(&a.tail_sums_c[m * 2 + 1], (double)s1);
      if constexpr (TAIL == 2) {
        atomic_add_d(&a.tail_sums_r[m * 2], (double)s0);
        atomic_add_d(&a.tail_sums_r[m * 2 + 1], (double)s2);
      }
    }
  }
  // ---- dW partial -> this workgroup's slab (plain stores; pw_bwd_wst.hip), or -> dw by fp32 atomics
  if (ww && !a.noflush) {
    float* slab = a.slab ? a.slab + (long long)blockIdx.x * a.Co * a.Ci : nullptr;
#pragma unroll
    for (int s = 0; s < NW; s++) {
      const int ci = s * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int co = wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (co < a.Co && ci < a.Ci) {
          if (slab) slab[(long long)co * a.Ci + ci] = acc_dw[s][e];
          else atomicAdd(&a.dw[(long long)co * a.Ci + ci], acc_dw[s][e]);
        }
      }
    }
  }
}

template <int NW, int KS, int CT>
static inline size_t bwa_lds_bytes() {
  return (size_t)2 * KS * 16 * 64 + (size_t)2 * CT * 32 * BWA_RP * 2 + (size_t)2 * NW * 32 * BWA_RP * 2 + (size_t)KS * 16 * 16 +
         (size_t)NW * 32 * WS_OP * 4;
}

// the layers this kernel is for: `a` convs with an identity shortcut whose dY tile is too tall for the sliced kernel --
// Co in 209..224 (fourteen k-steps), Ci <= 96: stage 4 of X3D-XS / S / M / L (96 <-> 216)
bool pw_bwd_wsta_applies(const x3d_pw_bwd_args* b) {
  if (x3d_env_int("X3D_PW_BWD_WSTA", 1) == 0) return false;   // A/B switch: 0 = off
  if (!x3d_is_half(b->dtype) || !b->w_panel || (!b->coef && !b->coef_fold) || !b->yraw || b->epi != X3D_EPI_ADD || !b->x || !b->add) return false;
  if (b->Cin <= 64 || b->Cin > 96 || ((b->Cout + 15) >> 4) != 14) return false;
  const long long P = (long long)b->T * b->H * b->W;
  if (P % 8 || P >= (1ll << 31) || (long long)b->Cin * P * 2 >= (1ll << 31)) return false;
  const void* ps[] = {b->g, b->yraw, b->dx, b->w_panel, b->x, b->add};
  for (const void* p : ps) if (!p || ((uintptr_t)p % 16)) return false;
  // (a shortcut conv below -- a third epilogue operand -- spills 76 bytes per lane: that one launch per model keeps its
  // separate x3d_tail_bwd, the caller falls back to the call without tail_c)
  if (b->tail_r) return false;
  if (b->tail_c && ((uintptr_t)b->tail_c % 16)) return false;
  return true;
}

template <typename H, int TAIL>
static int bwa_launch(PwBwdWstaArgs& a, hipStream_t st) {
  constexpr int NW = 3, KS = 14, CT = 7;
  const size_t lds = bwa_lds_bytes<NW, KS, CT>();
  X3D_DESCRIBE("pw_bwd_wsta_kernel<%s, %d, %d, %d, %d>", HV<H>::name, NW, KS, CT, TAIL);
  auto kern = pw_bwd_wsta_kernel<H, NW, KS, CT, TAIL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  const long long total_tiles = ceil_div_ll(a.P, 32) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_bwd_wsta: too many tiles");
  long long tpb, gx;
  x3d_persistent_grid(total_tiles, x3d_device_cus(), &tpb, &gx);
  a.tiles_per_block = (int)tpb;
  a.noflush = x3d_env_int("X3D_PW_BWD_NOFLUSH", 0);
  a.hot = x3d_env_int("X3D_PW_BWD_HOT", 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds, st, a);
  X3D_LAUNCH_CHECK("pw_bwd_wsta");
  return X3D_OK;
}

// called by x3d_pw_bwd (pw_bwd_fused.hip) when pw_bwd_wsta_applies()
int pw_bwd_wsta(const x3d_pw_bwd_args* b, hipStream_t st) {
  X3D_REQUIRE(!b->tail_c || x3d_describe.out || (b->tail_sums_c && (!b->tail_r || b->tail_sums_r)), "pw_bwd: tail_c / tail_r need their sums");
  PwBwdWstaArgs a;
  memset(&a, 0, sizeof(a));
  a.g = b->g; a.yraw = b->yraw; a.coef = b->coef;
  a.wp = b->w_panel; a.wp_rows = (b->Cin + 31) & ~31;
  a.dx = b->dx; a.add = b->add; a.x = b->x; a.dw = b->dw; a.slab = b->dw_slab; a.fold = bn_bwd_fold_arg(b->coef_fold);
  a.tail_c = b->tail_c; a.tail_r = b->tail_r; a.tail_sums_c = b->tail_sums_c; a.tail_sums_r = b->tail_sums_r;
  a.N = b->N; a.Co = b->Cout; a.Ci = b->Cin;
  a.P = (long long)b->T * b->H * b->W;
  const bool tail = b->tail_c != nullptr;
  if (b->dtype == X3D_F16) return tail ? bwa_launch<f16, 1>(a, st) : bwa_launch<f16, 0>(a, st);
  return tail ? bwa_launch<bf16, 1>(a, st) : bwa_launch<bf16, 0>(a, st);
}

int pw_bwd_wsta_dw_parts(const x3d_pw_bwd_args* b) {
  long long tpb, gx;
  x3d_persistent_grid(ceil_div_ll((long long)b->T * b->H * b->W, 32) * b->N, x3d_device_cus(), &tpb, &gx);
  return (int)gx;
}
