// Weights-stationary FUSED backward of the stage-4 `c` conv (216 inner channels <-> 96 block channels on 14 x 14 planes;
// reference model.py:292-299 through SURVEY appendix A): data gradient with the swish' epilogue + per-(n, c) sums AND the
// weight gradient, one pass over g / yraw / braw.
//
// The sliced fused kernel (pw_bwd_fused.hip) covers Ci = 216 with four 64-channel slices over blockIdx.y, each re-staging
// the dY tile (PMC traffic 1.4x of the algorithmic bytes), at 247 VGPRs and seven barriers per tile: 107 us per layer, the
// largest single item of the X3D-M backward pass (11 launches).  Here ONE persistent 8-wave workgroup per CU owns ALL input
// channels:
//   * dX = W^T dY as in pw_gemm_wst.h: wave w holds the 32 x Kp weight block of input channels 32w.. in REGISTERS
//     (KS x 4 VGPRs), the dY tile [Co][32 points] is staged once -- BN-backward prologue in fp32 -- into LDS in two
//     layouts: [k][32] for the transposed B-operand read of dX, and [co][32 + 8] for the row-wise A-operand read of dW;
//   * epilogue per wave through its private slab: u = gate * bn_b(braw), dv = dX * swish'(u) -> HBM, the per-(n, c) sums,
//     and Xh = swish(u) -> LDS rows 32w.. of the Xh tile;
//   * dW[co][ci] += dY[co][:] . Xh[ci][:]: wave w owns the CT tiles of ITS OWN input channels (cit = w), so the B operand
//     it reads is what it just wrote: no barrier between epilogue and weight-gradient MFMAs; CT x 16 accumulator VGPRs live
//     across all tiles of the workgroup, flushed once with fp32 atomics.
// One barrier per tile (the double-buffered dY tile).  g / yraw are prefetched two tiles ahead in registers, braw one.
#include <stdlib.h>

#include "pw_gemm_ws.h"

struct PwBwdWstArgs {
  const void* g; const void* yraw; const float* coef;   // dY = A*g + B*yraw + C   rows = Co
  const void* wp; int wp_rows;                          // dgrad panel (tiled image behind the row-major one)
  void* dx;                                             // [N][Ci][P]
  const void* braw; const float* b_ss; const float* egate; double* nc_sums;
  float* dw;                                            // [Co][Ci]
  int N, Co, Ci;
  long long P;
  int tiles_per_block;
  BnBwdFold fold;                                       // sums != NULL: the coefficient table is derived from the BatchNorm-backward sums here (x3d_hip.h coef_fold)
  float* slab;                                          // NULL | per-workgroup partial dW slabs [gridDim.x][Co][Ci] (plain stores)
  int hot;                                              // experiments build only (X3D_PW_BWD_HOT=1): every tile load re-reads the FIRST tile (cache hits): what memory latency costs
  int noflush;                                          // experiments build only (X3D_PW_BWD_NOFLUSH=1): timing without the dW flush
};

#define BW_YRP 40    // pitch (elements) of the row-read dY copy and of the Xh tile: 80 B = 5 units, odd -> b128 rows conflict-free

template <typename H, int NW, int KS, int CT>
__global__ __launch_bounds__(512, 2) void pw_bwd_wst_kernel(const PwBwdWstArgs a) {
  typedef typename HV<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BN = 32, OP = WS_OP, NT = 512, Kp = KS * 16, WP = Kp + 8, CoP = CT * 32, YRP = BW_YRP;
  static_assert(CoP >= Kp, "the row-read copy covers every staged dY row");
  constexpr int NSV = (Kp * 4 + NT - 1) / NT;            // dY staging vectors (8 points) per thread and tensor
  H* Yt = (H*)smem_raw;                                                          // [2][Kp][32]
  H* Yr = (H*)(smem_raw + (size_t)2 * Kp * 64);                                  // [2][CoP][YRP]
  H* Xh = (H*)(smem_raw + (size_t)2 * Kp * 64 + (size_t)2 * CoP * YRP * 2);      // [NW * 32][YRP]
  float* Cs = (float*)(smem_raw + (size_t)2 * Kp * 64 + (size_t)2 * CoP * YRP * 2 + (size_t)NW * 32 * YRP * 2);   // [Kp][4]
  float* Os = Cs + Kp * 4;                                                       // [NW][32][OP]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int mt = (a.Ci + 31) >> 5;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  if (tile_begin >= tile_end) return;                  // (never: the grid has no empty workgroup -- a slab would stay unwritten)
  float* myOs = Os + wid * 32 * OP;
  // input channels beyond NW row blocks: SLICES over blockIdx.y (stage 5: 432 = 2 x 7 row blocks), each slice a persistent
  // workgroup set of its own that re-stages the dY tile (K = Co rows) for its NW x 32 input channels
  const int rb = blockIdx.y * NW + wid;                // this wave's row block of input channels
  const bool mw = wid < NW && rb < mt;                 // ... if it owns one

  // ---- one-time set-up: zero the row-read copies' padding rows (Kp.. CoP), the BN-backward coefficient table
  {
    hx8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (H)0.f;
    for (int i = tid; i < 2 * CoP * YRP / 8; i += NT) ((hx8*)Yr)[i] = z;
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
  if (mw) {
    const H* wt = (const H*)a.wp + (long long)a.wp_rows * WP + ((long long)rb * KS * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) A[ks] = *(const hx8*)(wt + ks * 512);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int e = 0; e < 8; e++) A[ks][e] = (H)0.f;
  }

  // ---- dY staging: vector v = tid + NT * i -> row v >> 2, 8 points at unit v & 3 (unconditional clamped loads)
  hx8 g0[NSV], y0[NSV], g1[NSV], y1[NSV];
  auto issue_loads = [&](int tile_, hx8 (&gr)[NSV], hx8 (&yr)[NSV]) {
    const int tile = a.hot ? tile_begin : min(tile_, tile_end - 1);
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * NT;
      const int k = v >> 2;
      const long long p = p0 + (v & 3) * 8;
      const bool ok = k < a.Co && p < a.P;
      const long long o = ok ? ((long long)n * a.Co + k) * a.P + p : 0;
      gr[i] = *(const hx8*)((const T*)a.g + o);
      yr[i] = *(const hx8*)((const T*)a.yraw + o);
    }
  };
  auto commit = [&](int tile, int buf, const hx8 (&gr)[NSV], const hx8 (&yr)[NSV]) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = tid + i * NT;
      const int k = v >> 2;
      if (k >= Kp) continue;
      const bool ok = k < a.Co && p0 + (v & 3) * 8 < a.P;
      const f32x4 cf = *(const f32x4*)&Cs[k * 4];
      float val[8];
#pragma unroll
      for (int e = 0; e < 8; e++) val[e] = ok ? cf[0] * (float)gr[i][e] + cf[1] * (float)yr[i][e] + cf[2] : 0.f;   // (C must not leak into the padding)
      hx8 hv;
#pragma unroll
      for (int e = 0; e < 8; e++) hv[e] = (H)val[e];
      *(hx8*)&Yt[(buf * Kp + k) * BN + (v & 3) * 8] = hv;
      *(hx8*)&Yr[(buf * CoP + k) * YRP + (v & 3) * 8] = hv;
    }
  };

  // ---- braw prefetch (one tile ahead): lane -> row lane >> 1 of this wave's block, points 16 * (lane & 1) .. + 15
  const int row = lane >> 1, c0 = 16 * (lane & 1);
  const int m = rb * 32 + row;                         // this lane's input channel
  const bool mrow = mw && m < a.Ci;
  // Every load of the tile loop is UNCONDITIONAL (clamped address, value selected afterwards) and the SE gate travels with
  // braw: vmcnt retires in order, so a load issued behind the prefetches -- or a conditional one the compiler cannot
  // count -- turns the wait in front of the epilogue into vmcnt(0), i.e. a full memory latency per tile (r03h: 104 us)
  hx8 eb0[2], eb1[2];
  float gl0, gl1;
  const float* gsrc = a.egate ? a.egate : a.b_ss;      // (no SE: any valid address, the value is not used)
  auto issue_braw = [&](int tile_, hx8 (&eb)[2], float& gl) {
    const int tile = a.hot ? tile_begin : min(tile_, tile_end - 1);
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    gl = gsrc[(a.egate && mrow) ? (long long)n * a.Ci + m : 0];
#pragma unroll
    for (int hv = 0; hv < 2; hv++) {
      const long long p = p0 + c0 + 8 * hv;
      const long long o = (mrow && p < a.P) ? ((long long)n * a.Ci + m) * a.P + p : 0;
      eb[hv] = *(const hx8*)((const T*)a.braw + o);
    }
  };

  float st1 = 0.f, st2 = 0.f;                          // per-(sample, channel) sums of this lane's row
  auto flush_sums = [&](int n) {
    const float s1 = st1 + dpp_get<0xB1, 0xF>(st1), s2 = st2 + dpp_get<0xB1, 0xF>(st2);   // the two lanes of a row
    if (mrow && (lane & 1) == 0) {
      double* d = a.nc_sums + ((long long)n * a.Ci + m) * 2;
      atomic_add_d(d, (double)s1);
      atomic_add_d(d + 1, (double)s2);
    }
    st1 = 0.f;
    st2 = 0.f;
  };
  const float sb = mrow ? a.b_ss[m * 2] : 0.f, tb = mrow ? a.b_ss[m * 2 + 1] : 0.f;

  f32x16 acc_dw[CT];
#pragma unroll
  for (int s = 0; s < CT; s++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc_dw[s][e] = 0.f;

  const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_off = (8 * (g16 >> 1) + q) * BN + 16 * (g16 & 1) + 4 * pp;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

  int n_prev = tile_begin / tiles_per_n;
  issue_loads(tile_begin, g0, y0);
  issue_braw(tile_begin, eb0, gl0);
  __syncthreads();                                     // coefficient table, zeroed padding
  commit(tile_begin, 0, g0, y0);
  __syncthreads();
  issue_loads(tile_begin + 1, g1, y1);
  issue_loads(tile_begin + 2, g0, y0);

  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  auto step = [&](int tile, int cur, hx8 (&gr)[NSV], hx8 (&yr)[NSV], hx8 (&eb)[2], float gl, hx8 (&ebn)[2], float& gln) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    if (n != n_prev) flush_sums(n_prev);
    n_prev = n;
    issue_braw(tile + 1, ebn, gln);                    // a whole step to land
    const float gt = (mrow && a.egate) ? gl : 1.0f;
    // dx stores as bounds-checked buffer stores (offset past the sample's [Ci][P] matrix = dropped): a static count per tile
    __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc((T*)a.dx + (long long)n * a.Ci * a.P, 0,
                                                                   (int)((long long)a.Ci * a.P * 2), 0x00020000);

    // ---- dX = W^T dY: B operand from the current dY tile, A from registers
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    if (mw) {
      const H* xb = Yt + cur * (Kp * BN) + tr_off;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb + ks * 16 * BN));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xb + (ks * 16 + 4) * BN));
        const s16x8 bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc = mfma16<H>(A[ks], __builtin_bit_cast(hx8, bs), acc);
      }
    }

    // ---- the next dY tile goes into the other buffer (before the epilogue: its wait covers loads only)
    if (tile + 1 < tile_end) commit(tile + 1, cur ^ 1, gr, yr);

    if (mw) {
      // ---- epilogue through the wave-private slab
#pragma unroll
      for (int e = 0; e < 16; e++) myOs[((e & 3) + 8 * (e >> 2) + 4 * half) * OP + r] = acc[e];
#pragma unroll
      for (int hv = 0; hv < 2; hv++) {
        const long long p = p0 + c0 + 8 * hv;
        const bool ok = mrow && p < a.P;              // P % 8 == 0: a vector of 8 points is inside or outside
        float val[8];
        {
          const f32x4 v0 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv], v1 = *(const f32x4*)&myOs[row * OP + c0 + 8 * hv + 4];
#pragma unroll
          for (int e = 0; e < 4; e++) { val[e] = v0[e]; val[4 + e] = v1[e]; }
        }
        hx8 xh;
        const SwishCoef sc_ = swish_coef(sb, tb, gt);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float b = (float)eb[hv][e];
          float xs, d_;
          swish_bwd_(sc_, b, xs, d_);
          xh[e] = ok ? (H)xs : (H)0.f;                 // conv input of the forward pass (zero outside: dY is zero there too)
          const float dv = val[e] * d_;
          val[e] = dv;
          if (ok) { st1 += dv; st2 += dv * b; }
        }
        *(hx8*)&Xh[(wid * 32 + row) * YRP + c0 + 8 * hv] = xh;
        hx8 ov;
#pragma unroll
        for (int e = 0; e < 8; e++) ov[e] = (H)val[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, ov), dxr,
                                               ok ? (unsigned)(((long long)m * a.P + p) * 2) : 0x80000000u, 0, 0);
      }
      // ---- dW tiles (cot, cit = wid): A = dY rows (row-read copy), B = this wave's own Xh rows
      const H* yrow = Yr + (cur * CoP + r) * YRP + 8 * half;
      const H* xrow = Xh + (wid * 32 + r) * YRP + 8 * half;
#pragma unroll
      for (int ks = 0; ks < BN / 16; ks++) {
        const hx8 bf = *(const hx8*)(xrow + ks * 16);
#pragma unroll
        for (int s = 0; s < CT; s++) {
          const hx8 af = *(const hx8*)(yrow + s * 32 * YRP + ks * 16);
          acc_dw[s] = mfma16<H>(af, bf, acc_dw[s]);
        }
      }
    }
    __syncthreads();                                   // one barrier per tile
    issue_loads(tile + 3, gr, yr);
  };
  for (int tile = tile_begin; tile < tile_end; tile += 2) {
    step(tile, 0, g1, y1, eb0, gl0, eb1, gl1);
    if (tile + 1 < tile_end) step(tile + 1, 1, g0, y0, eb1, gl1, eb0, gl0);
  }
  flush_sums(n_prev);

  // ---- dW partial -> this workgroup's slab (plain stores, added up by a later launch: x3d_hip.h dw_slab), or -> dw by fp32
  // atomics.  256 workgroups x 83 KB of device-scope atomics take 17-24 us of this ~100 us launch whatever the schedule
  // (they execute at the memory side at ~1.3 TB/s chip-wide); the same bytes as stores ~4 us.
  if (mw && !a.noflush) {
    const int ci = rb * 32 + r;
    float* slab = a.slab ? a.slab + (long long)blockIdx.x * a.Co * a.Ci : nullptr;
#pragma unroll
    for (int s = 0; s < CT; s++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int co = s * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (co < a.Co && ci < a.Ci) {
          if (slab) slab[(long long)co * a.Ci + ci] = acc_dw[s][e];
          else atomicAdd(&a.dw[(long long)co * a.Ci + ci], acc_dw[s][e]);
        }
      }
  }
}

template <int NW, int KS, int CT>
static inline size_t bw_lds_bytes() {
  return (size_t)2 * KS * 16 * 64 + (size_t)2 * CT * 32 * BW_YRP * 2 + (size_t)NW * 32 * BW_YRP * 2 + (size_t)KS * 16 * 16 +
         (size_t)NW * 32 * WS_OP * 4;
}

// the layers this kernel is for: `c` convs whose input channels exceed one panel of the sliced kernel (129..224) with at
// most 96 output channels -- stage 4 of X3D-XS / S / M / L (216 <-> 96)
bool pw_bwd_wst_applies(const x3d_pw_bwd_args* b) {
  if (x3d_env_int("X3D_PW_BWD_WST", 1) == 0) return false;   // A/B switch: 0 = off
  if (!x3d_is_half(b->dtype) || !b->w_panel || (!b->coef && !b->coef_fold) || !b->yraw || b->epi != X3D_EPI_SWISH_BWD || b->tail_c) return false;
  // stage 4: 129..224 input channels, six k-steps (the panel's pitch is roundup(Co, 16) + 8); stage 5 (round 5): 225..448 input
  // channels as two slices of seven row blocks, twelve k-steps (432 <-> 192)
  const int ks = (b->Cout + 15) >> 4;
  if (!((b->Cin > 128 && b->Cin <= 224 && ks == 6) || (b->Cin > 224 && b->Cin <= 448 && ks == 12 && x3d_env_int("X3D_PW_BWD_WST5", 1) != 0))) return false;
  const long long P = (long long)b->T * b->H * b->W;
  if (P % 8 || P >= (1ll << 31) || (long long)b->Cin * P * 2 >= (1ll << 31)) return false;   // (2 GB buffer-store window per sample)
  const void* ps[] = {b->g, b->yraw, b->dx, b->w_panel, b->braw};
  for (const void* p : ps) if (!p || ((uintptr_t)p % 16)) return false;
  return true;
}

// slices over blockIdx.y and the persistent grid per slice (also what x3d_pw_bwd_dw_parts reports: one slab per blockIdx.x)
static inline void bw_grid(int N, long long P, int Ci, long long* tpb, long long* gx, int* slices) {
  *slices = ceil_div(ceil_div(Ci, 32), 7);
  x3d_persistent_grid(ceil_div_ll(P, 32) * N, x3d_device_cus() / *slices, tpb, gx);
}

template <typename H, int KS, int CT>
static int bw_launch(PwBwdWstArgs& a, hipStream_t st) {
  constexpr int NW = 7;
  const size_t lds = bw_lds_bytes<NW, KS, CT>();
  X3D_DESCRIBE("pw_bwd_wst_kernel<%s, %d, %d, %d>", HV<H>::name, NW, KS, CT);
  auto kern = pw_bwd_wst_kernel<H, NW, KS, CT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  const long long total_tiles = ceil_div_ll(a.P, 32) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_bwd_wst: too many tiles");
  long long tpb, gx;
  int slices;
  bw_grid(a.N, a.P, a.Ci, &tpb, &gx, &slices);
  a.tiles_per_block = (int)tpb;
  a.noflush = x3d_env_int("X3D_PW_BWD_NOFLUSH", 0);
  a.hot = x3d_env_int("X3D_PW_BWD_HOT", 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)slices), dim3(512), lds, st, a);
  X3D_LAUNCH_CHECK("pw_bwd_wst");
  return X3D_OK;
}

// called by x3d_pw_bwd (pw_bwd_fused.hip) when pw_bwd_wst_applies()
int pw_bwd_wst(const x3d_pw_bwd_args* b, hipStream_t st) {
  X3D_REQUIRE(b->braw && b->b_scale_shift && b->nc_sums, "pw_bwd: SWISH_BWD needs braw/b_scale_shift/nc_sums");
  PwBwdWstArgs a;
  memset(&a, 0, sizeof(a));
  a.g = b->g; a.yraw = b->yraw; a.coef = b->coef;
  a.wp = b->w_panel; a.wp_rows = (b->Cin + 31) & ~31;
  a.dx = b->dx; a.braw = b->braw; a.b_ss = b->b_scale_shift; a.egate = b->gate; a.nc_sums = b->nc_sums;
  a.dw = b->dw; a.slab = b->dw_slab; a.fold = bn_bwd_fold_arg(b->coef_fold);
  a.N = b->N; a.Co = b->Cout; a.Ci = b->Cin;
  a.P = (long long)b->T * b->H * b->W;
  if (((b->Cout + 15) >> 4) == 12) return b->dtype == X3D_F16 ? bw_launch<f16, 12, 6>(a, st) : bw_launch<bf16, 12, 6>(a, st);
  return b->dtype == X3D_F16 ? bw_launch<f16, 6, 3>(a, st) : bw_launch<bf16, 6, 3>(a, st);
}

// number of partial dW slabs a launch writes when given dw_slab (= its grid: x3d_hip.h)
int pw_bwd_wst_dw_parts(const x3d_pw_bwd_args* b) {
  long long tpb, gx;
  int slices;
  bw_grid(b->N, (long long)b->T * b->H * b->W, b->Cin, &tpb, &gx, &slices);
  return (int)gx;
}
