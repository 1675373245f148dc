// The `a` conv of a bottleneck WITHOUT its output tensor (reference model.py:305-309: a -> bn_a -> relu -> b).
//
// a_raw = W_a x is 2.25x the size of the block input x and, stored, is written once and read three times (depthwise
// forward, depthwise backward, `a`-conv backward).  It is a K = 24..48 GEMM on matrix cores that idle at < 5 %, so it is
// recomputed where it is consumed and never touches HBM:
//   x3d_pw_gram          one pass over x: the moment sums  [x ; 1] x^T  (Cin x Cin Gram matrix and sum x) on the matrix
//                        cores; with them the batch statistics of a_raw follow algebraically (a is linear in x):
//                        mean_c = w_c . sx / M,  E[a_c^2] = w_c^T XX w_c / M          (x3d_bn_finalize_gram)
//                        optional prologue = the folded residual tail / stem BatchNorm that builds x on load and stores it
//   x3d_ab_fwd           fused forward: a workgroup owns (sample, H-tile, 16 channels) and streams the T planes; per plane
//                        the x tile [Cin][rows x W] is staged in LDS as it lies in HBM, v_mfma_f32_16x16x32 (A = x^T read
//                        with ds_read_b64_tr_b16, B = 16 rows of W_a held in registers) produces the 16-channel plane of a,
//                        BN_a + ReLU are applied to the accumulators and the result lands in the fp32 plane image with zero
//                        halo that the 3x3x3 stencil reads (the stencil of dw_fwd.hip: three rotating partial planes per
//                        thread, packed FMAs) -- with the per-channel BN_b statistics and the SE pool in the epilogue.
// The same moment sums serve the `a`-conv backward (pw_bwd_rc.hip); the depthwise backward recomputes a the same way
// (x3d_ab_bwd, below).
#include <stdlib.h>

#include "dw_common.h"
#include "pw_gemm.h"

typedef __attribute__((ext_vector_type(4))) short s16x4_a;
typedef __attribute__((ext_vector_type(8))) short s16x8_a;
typedef s16x4_a __attribute__((address_space(3))) * lds_s16x4_a;

// ================================================================================================
// x3d_pw_gram
// ================================================================================================
struct GramArgs {
  const void* x;                 // [N][Ci][P]; with `raw`: written (the built input), else read
  const void* raw;               // prologue source 1 [N][Ci][P] (NULL: x is read as it is)
  const float* ss1;              // [Ci][2] scale / shift of raw
  const void* add;               // prologue source 2 [N][Ci][P] or NULL
  const float* ss2;              // [Ci][2] scale / shift of add (NULL: identity)
  double* gram;                  // [(Ci + 1)][Ci] +=   rows 0..Ci-1: x x^T, row Ci: sum x
  int N, Ci;
  long long P;
  int tiles_per_block;
};

#define GR_BN 128
#define GR_YP 160
#define GRAM_R 16

// RT: 32-row tiles of [x ; 1] (rows), CT: 32-row tiles of x (columns)
template <typename H, int RT, int CT, bool PRO>
__global__ __launch_bounds__(256, 6) void pw_gram_kernel(const GramArgs a) {
  typedef typename HV<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int BN = GR_BN, YP = GR_YP;
  constexpr int NT = RT * CT, TPW = (NT + 3) / 4, NKS = NT >= 4 ? 1 : 4 / NT;
  constexpr int NV = RT * 2;                        // staging vectors per thread (rows srow + 16 i)
  H* Zs = (H*)smem_raw;                             // [RT*32][YP], units swizzled by (row >> 2) & 3
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);
  const int srow = tid >> 4, sunit = tid & 15;

  hx8 rx[NV], ra[PRO ? NV : 1];
  float s1[PRO ? NV : 1], t1[PRO ? NV : 1], s2[PRO ? NV : 1], t2[PRO ? NV : 1];
  if constexpr (PRO) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int k = srow + 16 * i;
      const bool ok = k < a.Ci;
      s1[i] = ok ? a.ss1[k * 2] : 0.f; t1[i] = ok ? a.ss1[k * 2 + 1] : 0.f;
      s2[i] = (ok && a.add) ? (a.ss2 ? a.ss2[k * 2] : 1.f) : 0.f;
      t2[i] = (ok && a.add && a.ss2) ? a.ss2[k * 2 + 1] : 0.f;
    }
  }
  auto issue = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p = (long long)(tile - n * tiles_per_n) * BN + sunit * 8;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int k = srow + 16 * i;
      const long long o = (k < a.Ci && p < a.P) ? ((long long)n * a.Ci + k) * a.P + p : 0;
      rx[i] = *(const hx8*)((const H*)(PRO ? a.raw : a.x) + o);
      if constexpr (PRO) ra[i] = *(const hx8*)((const H*)(a.add ? a.add : a.raw) + o);
    }
  };
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((H*)a.x, 0, 0x7ffffff0, 0x00020000);
  auto commit = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p = (long long)(tile - n * tiles_per_n) * BN + sunit * 8;
    const bool pin = p < a.P;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int k = srow + 16 * i;
      hx8 v = rx[i];
      if constexpr (PRO) {     // x = relu(s1 * raw + t1 + (s2 * add + t2)), rounded to the storage type and stored
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = fmaxf(fmaf(s1[i], (float)rx[i][e], t1[i]) + fmaf(s2[i], (float)ra[i][e], t2[i]), 0.f);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (H)f[e];
        if constexpr (sizeof(long long) == 8) {
          const long long o = ((long long)n * a.Ci + k) * a.P + p;
          typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
          // (offsets beyond 2^31 bytes: per-sample resources would be needed; the host checks the tensor size)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), xr, (pin && k < a.Ci) ? (unsigned)(o * 2) : 0x80000000u, 0, 0);
        }
      }
      if (!(pin && k < a.Ci)) {
        const H fill = (pin && k == a.Ci) ? (H)1.f : (H)0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = fill;
      }
      *(hx8*)&Zs[k * YP + ((sunit ^ ((k >> 2) & 3)) << 3)] = v;
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

  if (tile_begin < tile_end) issue(tile_begin);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    __syncthreads();
    commit(tile);
    __syncthreads();
    issue(min(tile + 1, tile_end - 1));
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + 4 * s, kpart = 0;
      if constexpr (NKS > 1) { id = wid % NT; kpart = wid / NT; }
      if (id < NT && (NKS == 1 || kpart < NKS)) {
        const int rt = id / CT, ct = id - rt * CT;
        const int rowa = rt * 32 + r, rowb = ct * 32 + r;
        const int swa = (rowa >> 2) & 3, swb = (rowb >> 2) & 3;
        const H* arow = Zs + rowa * YP;
        const H* brow = Zs + rowb * YP;
        constexpr int KSTEPS = (BN / 16) / NKS;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
          const int u = (kpart * KSTEPS + ks) * 2 + half;
          const hx8 af = *(const hx8*)(arow + ((u ^ swa) << 3));
          const hx8 bf = *(const hx8*)(brow + ((u ^ swb) << 3));
          acc[s] = mfma16<H>(af, bf, acc[s]);
        }
      }
    }
  }
  if (tile_begin < tile_end) {
    // GRAM_R copies of the sums (as the BatchNorm statistics are replicated: every workgroup of the launch ends with the
    // same (Ci + 1) * Ci addresses -- with one copy 2048 workgroups queued on them: 187 -> 280 us on the 24-channel 112^2 input)
    double* gr = a.gram + (long long)(blockIdx.x % GRAM_R) * (a.Ci + 1) * a.Ci;
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + 4 * s;
      bool live = true;
      if constexpr (NKS > 1) { id = wid % NT; live = (wid / NT) < NKS; }
      if (id < NT && live) {
        const int rt = id / CT, ct = id - rt * CT;
        const int cj = ct * 32 + r;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int row = rt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
          if (row <= a.Ci && cj < a.Ci) atomic_add_d(&gr[(long long)row * a.Ci + cj], (double)acc[s][j]);
        }
      }
    }
  }
}

template <typename H, int RT, int CT>
static int gram_launch(GramArgs& a, hipStream_t st) {
  const size_t lds = (size_t)RT * 32 * GR_YP * 2;
  const bool pro = a.raw != nullptr;
  X3D_DESCRIBE("pw_gram_kernel<%s, %d, %d, %d>", HV<H>::name, RT, CT, (int)pro);
  static int slots[2] = {0, 0};
  if (slots[pro] == 0) {
    int dev = 0, cus = 256, nb = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    const void* kern = pro ? (const void*)pw_gram_kernel<H, RT, CT, true> : (const void*)pw_gram_kernel<H, RT, CT, false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 2;
    slots[pro] = nb * cus;      // a tile is two short barrier-separated phases: the latency hides behind the other workgroups of the CU
  }
  const long long total_tiles = ceil_div_ll(a.P, GR_BN) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_gram: too many tiles");
  long long tpb = ceil_div_ll(total_tiles, slots[pro]);
  if (tpb < 8) tpb = 8;
  a.tiles_per_block = (int)tpb;
  const dim3 grid((unsigned)ceil_div_ll(total_tiles, tpb));
  if (pro) hipLaunchKernelGGL((pw_gram_kernel<H, RT, CT, true>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((pw_gram_kernel<H, RT, CT, false>), grid, dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_gram");
  return X3D_OK;
}

static bool gram_supported(const x3d_pw_gram_args* g) {
  if (!g || !x3d_is_half(g->dtype) || g->Cin <= 0 || g->Cin > 64) return false;
  const long long P = (long long)g->T * g->H * g->W;
  if (P % 8 || P >= (1ll << 31)) return false;
  if ((long long)g->N * g->Cin * P * 2 >= (1ll << 31) && g->raw) return false;    // the built input is stored through one 2 GB window
  const void* ps[] = {g->x, g->raw, g->add};
  for (const void* p : ps) if (p && ((uintptr_t)p % 16)) return false;
  if ((g->raw && !g->raw_scale_shift) || (!g->raw && g->add)) return false;
  return g->x && g->gram;
}
extern "C" int x3d_pw_gram_supported(const x3d_pw_gram_args* g) { return gram_supported(g) ? 1 : 0; }
extern "C" long long x3d_pw_gram_elems(int Cin) { return Cin > 0 ? (long long)GRAM_R * (Cin + 1) * Cin : 0; }
extern "C" int x3d_pw_gram_replicas(void) { return GRAM_R; }

extern "C" int x3d_pw_gram(const x3d_pw_gram_args* g, void* stream) {
  X3D_REQUIRE(gram_supported(g), "pw_gram: shape / alignment / arguments not covered (x3d_pw_gram_supported() == 0)");
  GramArgs a;
  memset(&a, 0, sizeof(a));
  a.x = g->x; a.raw = g->raw; a.ss1 = g->raw_scale_shift; a.add = g->add; a.ss2 = g->add_scale_shift; a.gram = g->gram;
  a.N = g->N; a.Ci = g->Cin; a.P = (long long)g->T * g->H * g->W;
  hipStream_t st = (hipStream_t)stream;
  const int RT = ceil_div(g->Cin + 1, 32), CT = ceil_div(g->Cin, 32);
#define GR_CASE(H_, R_, C_) if (RT == R_ && CT == C_) return gram_launch<H_, R_, C_>(a, st);
  if (g->dtype == X3D_F16) { GR_CASE(f16, 1, 1) GR_CASE(f16, 2, 1) GR_CASE(f16, 2, 2) GR_CASE(f16, 3, 2) }
  else { GR_CASE(bf16, 1, 1) GR_CASE(bf16, 2, 1) GR_CASE(bf16, 2, 2) GR_CASE(bf16, 3, 2) }
#undef GR_CASE
  x3d_set_error("pw_gram: unsupported channel count");
  return X3D_ERR_INVALID;
}

// BatchNorm coefficients of a = W x from the moment sums of x (one thread per output channel)
template <typename H>
__global__ __launch_bounds__(64) void bn_finalize_gram_kernel(const double* __restrict__ gram, const float* __restrict__ w, x3d_bn_fold f,
                                                              int C, int Ci) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const long long rs = (long long)(Ci + 1) * Ci;
  double m1 = 0.0, m2 = 0.0;
  for (int i = 0; i < Ci; i++) {
    const double wi = (double)round_to<H>(w[(long long)c * Ci + i]);   // the matrix cores multiply the rounded weights
    double sxi = 0.0;
    for (int rp = 0; rp < GRAM_R; rp++) sxi += gram[rp * rs + (long long)Ci * Ci + i];
    m1 += wi * sxi;
    double q = 0.0;
    for (int j = 0; j < Ci; j++) {
      double gij = 0.0;
      for (int rp = 0; rp < GRAM_R; rp++) gij += gram[rp * rs + (long long)i * Ci + j];
      q += (double)round_to<H>(w[(long long)c * Ci + j]) * gij;
    }
    m2 += wi * q;
  }
  float sc, sh;
  bn_coefs(f, c, m1, m2, f.gamma[c], f.beta[c], true, sc, sh);
}

extern "C" int x3d_bn_finalize_gram(const double* gram, const float* w, double count, const float* gamma, const float* beta,
                                    float* moving_mean, float* moving_var, float eps, float momentum, int update_moving,
                                    float* scale_shift, float* mean_invstd, int C, int Cin, int dtype, void* stream) {
  X3D_REQUIRE(gram && w && gamma && beta && scale_shift && mean_invstd && C > 0 && Cin > 0 && count > 0, "bn_finalize_gram: bad args");
  X3D_REQUIRE(!update_moving || (moving_mean && moving_var), "bn_finalize_gram: moving stats required");
  X3D_REQUIRE(x3d_is_half(dtype), "bn_finalize_gram: 16-bit storage types only");
  x3d_bn_fold f;
  f.stats = nullptr; f.count = count; f.gamma = gamma; f.beta = beta; f.moving_mean = moving_mean; f.moving_var = moving_var;
  f.eps = eps; f.momentum = momentum; f.update_moving = update_moving; f.scale_shift = scale_shift; f.mean_invstd = mean_invstd;
  if (dtype == X3D_F16) hipLaunchKernelGGL(bn_finalize_gram_kernel<f16>, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, gram, w, f, C, Cin);
  else hipLaunchKernelGGL(bn_finalize_gram_kernel<bf16>, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, gram, w, f, C, Cin);
  X3D_LAUNCH_CHECK("bn_finalize_gram");
  return X3D_OK;
}

// ================================================================================================
// x3d_ab_fwd
// ================================================================================================
struct AbGeom {
  int N, C, Ci, T, H, W, Ho, Wo;
  int ph, pw;
  int TH, ntile_h, nstrips;      // output rows per tile, H-tiles, strips of SW outputs per row
  int RIN, LP;                   // staged input rows, plane pitch (floats)
  int NP, PX, NTILE;             // points of the x tile, its LDS pitch (elements), 16-point MFMA tiles
  int NG;                        // channel groups of 16
  int IPC;                       // stencil items (row, strip) per channel = TH * nstrips
  int units;                     // (n, H-tile) pairs
};
struct AbFwdArgs {
  AbGeom g;
  const void* x; const float* wa; const float* ss; const float* wb; void* y;
  double* stats; double* pool;
};

#define AB_CG 16
#define AB_THREADS 512

// a-plane producer: the 16-channel plane of relu(bn_a(W_a x)) for the staged x tile, written into the fp32 plane image
// (rows outside the image stay zero: the depthwise pads the ACTIVATION, not x).  Every lane of the workgroup takes part.
// A wave owns tiles wid, wid + nwaves, ... (TPWV of them, compile time): all transposed reads are issued first, then the
// MFMAs, then the BN + ReLU epilogues -- written as one loop per tile the LDS and matrix-core latencies of every tile were
// exposed one after the other.  dst[i] / vmask: this lane's plane offset per tile and which tiles lie in rows of the image
// (they do not depend on the plane: computed once per workgroup).
template <typename H, int KB, int TPWV>
__device__ __forceinline__ void ab_produce(const AbGeom& g, const H* __restrict__ xs, float* __restrict__ ap,
                                           const typename HV<H>::x8 (&bw)[KB], float sc, float sh, const int (&dst)[TPWV],
                                           unsigned vmask, int wid, int lane, int nwaves) {
  typedef typename HV<H>::x8 hx8;
  const int grp = lane >> 4, li = lane & 15, q = li >> 2, pq = li & 3;
  hx8 af[TPWV][KB];
#pragma unroll
  for (int i = 0; i < TPWV; i++) {
    int tl = wid + i * nwaves;
    if (tl >= g.NTILE) tl = g.NTILE - 1;        // (a wave without an i-th tile re-reads its last one: nothing is written)
    const int p0 = tl * 16;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      // lane 4q + p of a group supplies row (k block) + q, columns 4p .. 4p+3 of the 16-point block (T10); the k blocks past
      // Cin read rows that exist (their weights are zero)
      int kr = kb * 32 + grp * 8;
      if (kr + 8 > g.Ci) kr = 0;
      // image: row pitch = 64 * odd bytes (mod 256), and the 16-column blocks of rows 8..15 (mod 16) swapped in pairs: the two
      // 4-row blocks a 32-lane half reads (8 rows apart, same columns) then cover all 64 banks once
      const H* base = xs + (kr + q) * g.PX + (p0 ^ ((kr & 8) << 1)) + 4 * pq;
      const s16x4_a a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a)base);
      const s16x4_a a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a)(base + 4 * g.PX));
      const s16x8_a as = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      af[i][kb] = __builtin_bit_cast(hx8, as);
    }
  }
  f32x4 d[TPWV];
#pragma unroll
  for (int i = 0; i < TPWV; i++) {
    d[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      if constexpr (__is_same(H, bf16)) d[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kb], bw[kb], d[i], 0, 0, 0);
      else d[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][kb], bw[kb], d[i], 0, 0, 0);
    }
  }
  // D[m = point 4 grp + j][n = channel li]: four consecutive points of one row (W % 4 == 0)
#pragma unroll
  for (int i = 0; i < TPWV; i++) {
    if (dst[i] >= 0) {
      const bool in = (vmask >> i) & 1u;
      f32x4 u;
#pragma unroll
      for (int j = 0; j < 4; j++) u[j] = in ? fmaxf(fmaf(sc, d[i][j], sh), 0.f) : 0.f;
      float* o = ap + dst[i];
      if (g.pw == 0) *(f32x4*)o = u;
      else { o[0] = u[0]; o[1] = u[1]; o[2] = u[2]; o[3] = u[3]; }
    }
  }
}

// NX: x staging vectors (8 elements) per thread; KB: 32-wide k blocks of the `a` conv
template <typename H, int S, int SW, int NX, int KB, int TPWV>
__global__ __launch_bounds__(AB_THREADS, 4) void ab_fwd_kernel(const AbFwdArgs a) {
  typedef typename HV<H>::x8 hx8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const AbGeom& g = a.g;
  constexpr int WIN = (SW - 1) * S + 3;
  constexpr int EB = 2;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float* ap = (float*)smem_raw;                                         // [16][RIN][LP]
  const int ap_floats = AB_CG * g.RIN * g.LP;
  H* xs = (H*)(smem_raw + (size_t)ap_floats * 4);                       // [Ci][PX]

  // workgroup -> (unit = (n, H-tile), channel group); the NG groups of a unit sit on ONE XCD (blockIdx mod 8) and are
  // dispatched together, so the x tile they all stage is fetched from HBM once
  const int b = blockIdx.x;
  const int xcd = b & 7, qd = b >> 3;
  const int cg = qd % g.NG, unit = (qd / g.NG) * 8 + xcd;
  if (unit >= g.units) return;
  const int tile = unit % g.ntile_h, n = unit / g.ntile_h;
  const int h0 = tile * g.TH;
  const int row0 = h0 * S - g.ph;

  for (int i = tid; i < ap_floats; i += AB_THREADS) ap[i] = 0.f;

  // ---- matrix-core operands of this lane: 8 consecutive input channels of weight row cg*16 + (lane & 15), BN_a of that row
  hx8 bw[KB];
  float sc = 0.f, sh = 0.f;
  {
    const int c = cg * AB_CG + (lane & 15);
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = kb * 32 + (lane >> 4) * 8 + j;
        bw[kb][j] = (H)((c < g.C && k < g.Ci) ? a.wa[(long long)c * g.Ci + k] : 0.f);
      }
    }
    if (c < g.C) { sc = a.ss[c * 2]; sh = a.ss[c * 2 + 1]; }
  }

  // ---- this lane's destinations in the plane image, one per matrix-core tile of its wave
  int dst[TPWV];
  unsigned vmask = 0;
#pragma unroll
  for (int i = 0; i < TPWV; i++) {
    const int tl = wid + i * (AB_THREADS / 64);
    const int p = tl * 16 + 4 * (lane >> 4);
    dst[i] = -1;
    if (tl < g.NTILE && p < g.NP) {
      const int lr = p / g.W, w0 = p - lr * g.W;
      const int hi = row0 + lr;
      dst[i] = ((lane & 15) * g.RIN + lr) * g.LP + g.pw + w0;
      if (hi >= 0 && hi < g.H) vmask |= 1u << i;
    }
  }

  // ---- stencil item of this thread: (channel, output row, strip)
  const int ch = tid / g.IPC, rem = tid - ch * g.IPC;
  const int r = rem / g.nstrips, sidx = rem - r * g.nstrips;
  const int c_it = cg * AB_CG + ch;
  const int ho = h0 + r, wo0 = sidx * SW;
  const bool active = ch < AB_CG && c_it < g.C && ho < g.Ho;
  v2f w21[3][3];
  float w0_[3][3];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const float* wp = a.wb + (long long)(active ? c_it : 0) * 27;
    w21[k / 3][k % 3] = (v2f){wp[18 + k], wp[9 + k]};
    w0_[k / 3][k % 3] = wp[k];
  }

  // ---- x staging map: vector v of the tile = (input channel k, 8 consecutive points); whole rows are contiguous in memory
  const int nvk = g.NP / 8;                         // vectors per input channel
  const long long iplane = (long long)g.H * g.W;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
      (H*)a.x + (long long)n * g.Ci * g.T * iplane, 0, (int)((long long)g.Ci * g.T * iplane * EB), 0x00020000);
  int gx[NX], lx[NX];
#pragma unroll
  for (int i = 0; i < NX; i++) {
    const int v = tid + i * AB_THREADS;
    gx[i] = DW_OOB; lx[i] = -1;
    if (v < nvk * g.Ci) {
      const int k = v / nvk, pv = (v - k * nvk) * 8;
      const int lr = pv / g.W, w = pv - lr * g.W;      // W % 8 == 0: a vector stays inside its row
      const int hi = row0 + lr;
      lx[i] = k * g.PX + (pv ^ ((k & 8) << 1));          // (the block flip of the image: ab_produce)
      if (hi >= 0 && hi < g.H) gx[i] = (int)(((long long)k * g.T * iplane + (long long)hi * g.W + w) * EB);
    }
  }
  Raw rawx[NX];
  const int iplB = (int)(iplane * EB);
#pragma unroll
  for (int i = 0; i < NX; i++) raw_bload<16>(rawx[i], rsX, gx[i], 0);

  const long long oplane = (long long)g.Ho * g.Wo;
  // (one resource per sample, the channel in the offset: a per-thread resource would be a waterfall loop per access)
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      (H*)a.y + (long long)n * g.C * g.T * oplane, 0, (int)((long long)g.C * g.T * oplane * EB), 0x00020000);
  const int oplB = (int)(oplane * EB);
  const int oY = active ? (int)(((long long)c_it * g.T * oplane + (long long)ho * g.Wo + wo0) * EB) : DW_OOB;

  v2f acc01[SW], acc2p[(SW + 1) / 2];
  float fin[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) { acc01[i] = (v2f){0.f, 0.f}; fin[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < (SW + 1) / 2; j++) acc2p[j] = (v2f){0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  auto store_plane = [&](int t, bool live, const float (&v)[SW]) {
    Raw o;
    raw_pack<H, SW>(o, v);
    raw_bstore<SW * EB>(o, rsY, live ? oY + t * oplB : DW_OOB, 0);
#pragma unroll
    for (int i = 0; i < SW; i++) {
      const float u = (live && active) ? v[i] : 0.f;
      s1 += u;
      s2 += u * u;
    }
  };

  for (int t = 0; t < g.T; ++t) {
    // (1) the staged x plane into LDS (as it lies in memory: the storage type is the matrix-core operand type)
#pragma unroll
    for (int i = 0; i < NX; i++) {
      if (lx[i] >= 0) {
        const dw_u32x4 v = {rawx[i].w[0], rawx[i].w[1], rawx[i].w[2], rawx[i].w[3]};
        *(dw_u32x4*)(xs + lx[i]) = v;
      }
    }
    __syncthreads();     // x tile complete; every thread has left the stencil of plane t-1
    // (2) next plane's loads fly under the matrix-core and stencil phases (past T: out of range, nothing moves)
#pragma unroll
    for (int i = 0; i < NX; i++) raw_bload<16>(rawx[i], rsX, gx[i] + (t + 1 < g.T ? (t + 1) * iplB : DW_OOB), 0);
    // (3) the a plane of this workgroup's 16 channels
    ab_produce<H, KB, TPWV>(g, xs, ap, bw, sc, sh, dst, vmask, wid, lane, AB_THREADS / 64);
    __syncthreads();
    // (4) stencil: this plane feeds output planes t-1 (kt = 2), t (kt = 1), t+1 (kt = 0)
    store_plane(t - 2, t >= 2, fin);
    if (active) {
#pragma unroll
      for (int kh = 0; kh < 3; kh++) {
        float win[WIN];
        const float* row = ap + ((ch * g.RIN) + r * S + kh) * g.LP + wo0 * S;
        lds_window<WIN, (SW * S >= 4 ? 4 : SW * S)>(row, win);
        dw_taps_row<S, SW, WIN>(win, w21[kh], w0_[kh], acc01, acc2p);
      }
    }
    dw_rotate<SW>(fin, acc01, acc2p);
  }
  store_plane(g.T - 2, g.T >= 2, fin);
  float last[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) last[i] = acc01[i].x;
  store_plane(g.T - 1, true, last);

  // ---- per-channel sums of this workgroup: items of one channel are consecutive threads; fixed-order reduction through LDS
  if (a.stats || a.pool) {
    __syncthreads();
    float* red = ap;                                 // [threads][2]
    red[tid * 2] = s1; red[tid * 2 + 1] = s2;
    __syncthreads();
    if (tid < AB_CG) {
      const int c = cg * AB_CG + tid;
      if (c < g.C) {
        float q1 = 0.f, q2 = 0.f;
        for (int i = 0; i < g.IPC; i++) { q1 += red[(tid * g.IPC + i) * 2]; q2 += red[(tid * g.IPC + i) * 2 + 1]; }
        if (a.stats) {
          double* sp = stats_replica(a.stats, g.C, (unsigned)unit);
          atomic_add_d(&sp[c * 2], (double)q1);
          atomic_add_d(&sp[c * 2 + 1], (double)q2);
        }
        if (a.pool) atomic_add_d(&a.pool[(long long)n * g.C + c], (double)q1);
      }
    }
  }
}

static bool ab_geom(AbGeom& g, int N, int Cin, int C, int T, int H, int W, int S, int SW) {
  g.N = N; g.C = C; g.Ci = Cin; g.T = T; g.H = H; g.W = W;
  g.Ho = ceil_div(H, S); g.Wo = ceil_div(W, S);
  const int tot_h = (g.Ho - 1) * S + 3 - H, tot_w = (g.Wo - 1) * S + 3 - W;
  g.ph = (tot_h > 0 ? tot_h : 0) / 2;
  g.pw = (tot_w > 0 ? tot_w : 0) / 2;
  if (g.Wo % SW) return false;
  g.nstrips = g.Wo / SW;
  if (g.nstrips > AB_THREADS / AB_CG) return false;
  int th = (AB_THREADS / AB_CG) / g.nstrips;
  if (th > g.Ho) th = g.Ho;
  g.ntile_h = ceil_div(g.Ho, th);
  g.TH = ceil_div(g.Ho, g.ntile_h);
  g.IPC = g.TH * g.nstrips;
  g.RIN = (g.TH - 1) * S + 3;
  g.LP = ((g.nstrips * SW - 1) * S + 3 + 3) & ~3;
  g.NP = g.RIN * W;
  g.NTILE = ceil_div(g.NP, 16);
  int px = g.NTILE * 16;                                 // >= NP, and the transposed reads of the last tile stay inside a row
  px = ((px - 32 + 63) / 64) * 64 + 32;                  // pitch = 64 * odd bytes (mod 256), an even number of 16-column blocks
  g.PX = px;
  g.NG = ceil_div(C, AB_CG);
  g.units = N * g.ntile_h;
  return true;
}
static size_t ab_fwd_lds(const AbGeom& g) {
  size_t a = (size_t)AB_CG * g.RIN * g.LP * 4;
  if (a < (size_t)AB_THREADS * 2 * 4) a = (size_t)AB_THREADS * 2 * 4;
  return a + (size_t)g.Ci * g.PX * 2 + 64;
}

static bool ab_fwd_shape(const x3d_ab_fwd_args* f, AbGeom* g, int* SW, int* NX) {
  if (!f || !x3d_is_half(f->dtype) || (f->stride != 1 && f->stride != 2)) return false;
  if (f->Cin <= 0 || f->Cin > 64 || (f->Cin % 8) || f->C <= 0 || f->T <= 0 || f->H <= 0 || f->W <= 0 || (f->W % 8)) return false;
  const int wo = ceil_div(f->W, f->stride);
  *SW = (wo % 4 == 0 && wo >= 20) ? 4 : ((wo % 2 == 0) ? 2 : 0);
  if (*SW == 0 || !ab_geom(*g, f->N, f->Cin, f->C, f->T, f->H, f->W, f->stride, *SW)) return false;
  *NX = ceil_div(g->Ci * (g->NP / 8), AB_THREADS);
  if (*NX > 6 || (g->NP % 8) || ceil_div(g->NTILE, AB_THREADS / 64) > 8) return false;
  if (ab_fwd_lds(*g) > 80 * 1024) return false;                     // two workgroups per CU
  const long long per_n = (long long)f->Cin * f->T * f->H * f->W * 2, per_o = (long long)f->C * f->T * g->Ho * g->Wo * 2;
  if (per_n >= DW_OOB || per_o >= DW_OOB) return false;
  if (((uintptr_t)f->x % 16) || ((uintptr_t)f->y % (2 * *SW))) return false;
  return f->x && f->a_w && f->a_scale_shift && f->b_w && f->y;
}
extern "C" int x3d_ab_fwd_supported(const x3d_ab_fwd_args* f) {
  AbGeom g; int sw, nx;
  return ab_fwd_shape(f, &g, &sw, &nx) ? 1 : 0;
}

// staging vectors per thread and matrix-core tiles per wave are compile-time: rounded up to the instantiated sizes
static inline int ab_round_nx(int nx) { return nx <= 2 ? 2 : (nx <= 4 ? 4 : 6); }
static inline int ab_round_tp(int tp) { return tp <= 2 ? 2 : (tp <= 3 ? 3 : (tp <= 5 ? 5 : 8)); }

template <typename H, int S, int SW, int KB>
static int ab_fwd_launch(const AbFwdArgs& a, int NX, hipStream_t st) {
  const int nx = ab_round_nx(NX), tp = ab_round_tp(ceil_div(a.g.NTILE, AB_THREADS / 64));
  X3D_DESCRIBE("ab_fwd_kernel<%s, %d, %d, %d, %d, %d>", HV<H>::name, S, SW, nx, KB, tp);
  const size_t lds = ab_fwd_lds(a.g);
  const unsigned grid = (unsigned)(ceil_div(a.g.units, 8) * 8 * a.g.NG);
#define AB_INST(N_, T_)                                                                                   \
  if (nx == N_ && tp == T_) {                                                                             \
    auto kern = ab_fwd_kernel<H, S, SW, N_, KB, T_>;                                                      \
    static bool attr_set = false;                                                                         \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(AB_THREADS), lds, st, a);                                   \
    X3D_LAUNCH_CHECK("ab_fwd");                                                                           \
    return X3D_OK;                                                                                        \
  }
  AB_INST(2, 2) AB_INST(2, 3) AB_INST(2, 5) AB_INST(2, 8) AB_INST(4, 2) AB_INST(4, 3) AB_INST(4, 5) AB_INST(4, 8)
  AB_INST(6, 3) AB_INST(6, 5) AB_INST(6, 8)
#undef AB_INST
  x3d_set_error("ab_fwd: no instantiation for %d staging vectors / %d tiles per wave", nx, tp);
  return X3D_ERR_INVALID;
}

extern "C" int x3d_ab_fwd(const x3d_ab_fwd_args* f, void* stream) {
  AbFwdArgs a;
  int SW = 0, NX = 0;
  X3D_REQUIRE(ab_fwd_shape(f, &a.g, &SW, &NX), "ab_fwd: shape / alignment not covered (x3d_ab_fwd_supported() == 0)");
  a.x = f->x; a.wa = f->a_w; a.ss = f->a_scale_shift; a.wb = f->b_w; a.y = f->y; a.stats = f->stats; a.pool = f->pool;
  hipStream_t st = (hipStream_t)stream;
  const int KB = ceil_div(f->Cin, 32);
#define AB_CASE(H_, S_, W_, K_) if (f->stride == S_ && SW == W_ && KB == K_) return ab_fwd_launch<H_, S_, W_, K_>(a, NX, st);
  if (f->dtype == X3D_F16) {
    AB_CASE(f16, 2, 4, 1) AB_CASE(f16, 2, 2, 1) AB_CASE(f16, 1, 4, 1) AB_CASE(f16, 1, 2, 1)
    AB_CASE(f16, 2, 4, 2) AB_CASE(f16, 2, 2, 2) AB_CASE(f16, 1, 4, 2) AB_CASE(f16, 1, 2, 2)
  } else {
    AB_CASE(bf16, 2, 4, 1) AB_CASE(bf16, 2, 2, 1) AB_CASE(bf16, 1, 4, 1) AB_CASE(bf16, 1, 2, 1)
    AB_CASE(bf16, 2, 4, 2) AB_CASE(bf16, 2, 2, 2) AB_CASE(bf16, 1, 4, 2) AB_CASE(bf16, 1, 2, 2)
  }
#undef AB_CASE
  x3d_set_error("ab_fwd: no instantiation");
  return X3D_ERR_INVALID;
}
