// x3d_pw_wgrad: pointwise convolution weight gradient on fp32-input MFMA
#include "common.h"

// ------------------------------------------------------------------------------------------------
// weight gradient: split over points, fp32 atomics into dw
// ------------------------------------------------------------------------------------------------
struct PwWgradArgs {
  const void* g; const void* yraw; const float* coef;                 // dY operand (rows = Cout)
  const void* x; const float* xcoef; const float* xgate; int xact;   // X operand (rows = Cin)
  float* dw;
  int N, Cout, Cin;
  long long P, Pin;
  int stride, H, W, Ho, Wo;
  int mt_per_group;     // 32-row tiles of Cout handled by one blockIdx.y
  int nt_total;         // 32-col tiles of Cin
  int steps_per_block;  // 32-point steps per block
  int noflush;          // X3D_PW_WG_NOFLUSH=1 (timing experiment only: the partial tiles are NOT added to dw)
  int ragged;           // 16-bit storage, P % 8 != 0, stride 1: the vector kernel with ragged row ends (pw_gemm.h)
  BnBwdFold fold;       // sums != NULL: the dY coefficients are derived from the BatchNorm-backward sums (x3d_hip.h coef_fold; pw_wgrad_bf16_v2 only)
  float* slab;          // NULL | partial weight gradients [gridDim.x][Cout][Cin], plain stores (x3d_hip.h dw_slab)
  int slab_parts;       // slabs the buffer holds: the launcher refuses another grid (host side only)
};

#include "pw_wgrad_bf16.h"
#include "pw_wgrad_f32p.h"

template <typename T, int VEC, int TPW, bool XPRO, bool STRIDED>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const PwWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BP = 32, LP = 33;  // points per step, LDS pitch (odd: conflict-free row-strided reads)
  const int rowsA = a.mt_per_group * 32, rowsB = a.nt_total * 32;
  float* As = smem;                // [rowsA][LP]
  float* Bs = smem + rowsA * LP;   // [rowsB][LP]
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

  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[s][j] = 0.f;

  constexpr int VPR = BP / VEC;
  // per-row coefficients in LDS, once per workgroup: read from global next to every staged vector they were three (A rows) /
  // two or three (B rows) dependent loads in front of each LDS write
  float* Ca = Bs + rowsB * LP;     // [rowsA][4]  {A, B, C} of dY = A g + B yraw + C (zeros without coef)
  float* Cb = Ca + rowsA * 4;      // [rowsB][4]  {s, t, gate} of the X prologue
  for (int row = tid; row < rowsA; row += 256) {
    const int co = co0 + row;
    const bool ok = co < a.Cout && a.coef;
    Ca[row * 4] = ok ? a.coef[co * 4] : 1.f; Ca[row * 4 + 1] = ok ? a.coef[co * 4 + 1] : 0.f; Ca[row * 4 + 2] = ok ? a.coef[co * 4 + 2] : 0.f;
  }
  if constexpr (XPRO) {
    for (int row = tid; row < rowsB; row += 256) {
      const bool ok = row < a.Cin;
      Cb[row * 4] = ok ? a.xcoef[row * 2] : 0.f; Cb[row * 4 + 1] = ok ? a.xcoef[row * 2 + 1] : 0.f;
      Cb[row * 4 + 2] = (ok && a.xgate) ? a.xgate[(long long)n * a.Cin + row] : 1.0f;
    }
  }
  constexpr int UX = 4;            // vectors per thread and round, their loads issued together (the rolled loop waited for each in turn)
  for (int step = s_begin; step < s_end; ++step) {
    const long long p0 = (long long)step * BP;
    __syncthreads();
    for (int base = 0; base < rowsA * VPR; base += 256 * UX) {
      float gv[UX][VEC], yv[UX][VEC];
#pragma unroll
      for (int u = 0; u < UX; u++) {
        const int v = base + u * 256 + tid;
        const int row = v / VPR, pv = v - row * VPR;
        const int co = co0 + row;
        const long long p = p0 + (long long)pv * VEC;
#pragma unroll
        for (int e = 0; e < VEC; e++) { gv[u][e] = 0.f; yv[u][e] = 0.f; }
        if (v < rowsA * VPR && co < a.Cout && p < a.P) {
          const long long o = ((long long)n * a.Cout + co) * a.P + p;
          VecIO<T, VEC>::load((const T*)a.g + o, gv[u]);
          if (a.coef) VecIO<T, VEC>::load((const T*)a.yraw + o, yv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UX; u++) {
        const int v = base + u * 256 + tid;
        if (v >= rowsA * VPR) continue;
        const int row = v / VPR, pv = v - row * VPR;
        const long long p = p0 + (long long)pv * VEC;
        const bool ok = co0 + row < a.Cout && p < a.P;
        const float A = Ca[row * 4], B = Ca[row * 4 + 1], C = Ca[row * 4 + 2];
#pragma unroll
        for (int e = 0; e < VEC; e++) As[row * LP + pv * VEC + e] = ok ? (A * gv[u][e] + B * yv[u][e] + C) : 0.f;
      }
    }
    for (int base = 0; base < rowsB * VPR; base += 256 * UX) {
      float xv[UX][VEC];
#pragma unroll
      for (int u = 0; u < UX; u++) {
        const int v = base + u * 256 + tid;
        const int row = v / VPR, pv = v - row * VPR;
        const long long p = p0 + (long long)pv * VEC;
#pragma unroll
        for (int e = 0; e < VEC; e++) xv[u][e] = 0.f;
        if (v < rowsB * VPR && row < a.Cin && p < a.P) {
          if constexpr (STRIDED) {
            const long long hw = (long long)a.Ho * a.Wo;
            const long long t = p / hw;
            const int rem = (int)(p - t * hw);
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const long long src = (t * a.H + (long long)ho * a.stride) * a.W + (long long)wo * a.stride;
            xv[u][0] = to_f<T>(((const T*)a.x)[((long long)n * a.Cin + row) * a.Pin + src]);
          } else {
            VecIO<T, VEC>::load((const T*)a.x + ((long long)n * a.Cin + row) * a.Pin + p, xv[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UX; u++) {
        const int v = base + u * 256 + tid;
        if (v >= rowsB * VPR) continue;
        const int row = v / VPR, pv = v - row * VPR;
        const long long p = p0 + (long long)pv * VEC;
        const bool ok = row < a.Cin && p < a.P;
        float val[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) val[e] = xv[u][e];
        if constexpr (XPRO) {
          const float s_ = Cb[row * 4], t_ = Cb[row * 4 + 1], g_ = Cb[row * 4 + 2];
#pragma unroll
          for (int e = 0; e < VEC; e++) val[e] = (s_ * val[e] + t_) * g_;
          act_vec<VEC>(val, a.xact);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) Bs[row * LP + pv * VEC + e] = ok ? val[e] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int id = wid + 4 * s;
      if (id < ntiles) {
        const int mt = id / a.nt_total, nt = id - mt * a.nt_total;
        const float* ap = As + (mt * 32 + r) * LP + half;
        const float* bp = Bs + (nt * 32 + r) * LP + half;
#pragma unroll
        for (int kk = 0; kk < BP; kk += 2)
          acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc[s], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int s = 0; s < TPW; s++) {
    const int id = wid + 4 * s;
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

template <typename T, int VEC, int TPW, bool XPRO, bool STRIDED>
static int pw_wgrad_launch(PwWgradArgs& a, hipStream_t st) {
  const int mt_total = ceil_div(a.Cout, 32);
  a.nt_total = ceil_div(a.Cin, 32);
  int g = (4 * TPW) / a.nt_total;
  if (g < 1) g = 1;
  if (g > mt_total) g = mt_total;
  a.mt_per_group = g;
  const int gy = ceil_div(mt_total, g);
  const long long steps_per_n = ceil_div_ll(a.P, 32);
  long long total = steps_per_n * a.N;
  int spb = (int)(total / 1024);
  if (spb < 4) spb = 4;
  if (spb > 64) spb = 64;
  if (spb > steps_per_n) spb = (int)steps_per_n;
  a.steps_per_block = spb;
  const long long gx = ceil_div_ll(steps_per_n, spb) * a.N;
  const size_t lds = (size_t)(a.mt_per_group + a.nt_total) * 32 * (33 + 4) * sizeof(float);   // tiles + the coefficient tables
  X3D_DESCRIBE("pw_wgrad_kernel<float, %d, %d, %d, %d>", VEC, TPW, (int)XPRO, (int)STRIDED);
  auto kern = pw_wgrad_kernel<T, VEC, TPW, XPRO, STRIDED>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_wgrad");
  return X3D_OK;
}

template <typename T, int VEC, bool XPRO, bool STRIDED>
static int pw_wgrad_tpw(PwWgradArgs& a, hipStream_t st) {
  const int nt = ceil_div(a.Cin, 32), mt = ceil_div(a.Cout, 32);
  const int tiles = nt * mt;
  if (tiles <= 4) return pw_wgrad_launch<T, VEC, 1, XPRO, STRIDED>(a, st);
  if (tiles <= 8) return pw_wgrad_launch<T, VEC, 2, XPRO, STRIDED>(a, st);
  if (tiles <= 16) return pw_wgrad_launch<T, VEC, 4, XPRO, STRIDED>(a, st);
  return pw_wgrad_launch<T, VEC, 8, XPRO, STRIDED>(a, st);  // nt <= 32 checked by the caller
}

template <typename T>
static int pw_wgrad_dispatch(PwWgradArgs& a, int vec, bool xpro, hipStream_t st) {
  constexpr int FULL = 16 / sizeof(T);
  if (a.stride > 1) {
    if (xpro) { x3d_set_error("pw_wgrad: strided input takes no prologue"); return X3D_ERR_INVALID; }
    return pw_wgrad_tpw<T, 1, false, true>(a, st);
  }
  if (vec >= FULL)
    return xpro ? pw_wgrad_tpw<T, FULL, true, false>(a, st) : pw_wgrad_tpw<T, FULL, false, false>(a, st);
  return xpro ? pw_wgrad_tpw<T, 1, true, false>(a, st) : pw_wgrad_tpw<T, 1, false, false>(a, st);
}

extern "C" int x3d_pw_wgrad_dw_parts(const x3d_pw_wgrad_args* w);

extern "C" int x3d_pw_wgrad(const x3d_pw_wgrad_args* w, void* stream) {
  X3D_REQUIRE(w && w->g && w->x && w->dw, "pw_wgrad: null pointer");
  X3D_REQUIRE(((w->coef == nullptr) && (w->coef_fold == nullptr)) == (w->yraw == nullptr), "pw_wgrad: coef (or coef_fold) and yraw go together");
  // (dry runs -- kernel name, slab-part count -- are asked while a plan is still being recorded: its accumulators, the fold's
  // `sums` among them, have no address yet)
  X3D_REQUIRE(x3d_describe.out || x3d_parts_query || (bn_bwd_fold_ok(w->coef_fold) && !(w->coef_fold && w->coef_fold->dgamma)),
              "pw_wgrad: bad coef_fold (the weight-gradient launch never publishes dgamma / dbeta)");
  X3D_REQUIRE(w->stride == 1 || w->stride == 2, "pw_wgrad: stride must be 1 or 2");
  X3D_REQUIRE(w->N > 0 && w->Cin > 0 && w->Cout > 0 && w->T > 0 && w->H > 0 && w->W > 0,
              "pw_wgrad: bad extents");
  X3D_REQUIRE(x3d_dtype_ok(w->dtype), "pw_wgrad: bad dtype");
  X3D_REQUIRE(w->Cin <= 32 * 32, "pw_wgrad: Cin too large");
  PwWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.g = w->g; a.yraw = w->yraw; a.coef = w->coef; a.fold = bn_bwd_fold_arg(w->coef_fold);
  a.x = w->x; a.xcoef = w->in_scale_shift; a.xgate = w->in_gate; a.xact = w->in_act;
  a.dw = w->dw; a.slab = w->dw_slab; a.slab_parts = w->dw_slab_parts; a.N = w->N; a.Cout = w->Cout; a.Cin = w->Cin;
  a.stride = w->stride; a.H = w->H; a.W = w->W;
  a.Ho = ceil_div(w->H, w->stride); a.Wo = ceil_div(w->W, w->stride);
  a.Pin = (long long)w->T * w->H * w->W;
  a.P = (long long)w->T * a.Ho * a.Wo;
  X3D_REQUIRE(a.Pin < (1ll << 31) && a.P < (1ll << 31), "pw_wgrad: more than 2^31 points per sample");   // 32-bit point indices in the kernels
  const bool xpro = w->in_scale_shift != nullptr;
  X3D_REQUIRE(xpro || (!w->in_gate && w->in_act == X3D_ACT_NONE), "pw_wgrad: prologue needs in_scale_shift");
  const int eb = w->dtype == X3D_F32 ? 4 : 2;
  const int vec = pick_vec(eb, a.P, w->g, w->yraw, w->x);
  a.ragged = (pw_ragged_rows(a.P, eb) &&
              (((uintptr_t)w->g | (uintptr_t)w->yraw | (uintptr_t)w->x) % 16) == 0) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  // dw_slab: only where the kernel behind the call has the form (x3d_pw_wgrad_dw_parts() > 0).  The launcher that is chosen
  // checks it -- a kernel without the form refuses a slab, one with it refuses a buffer whose slab count (dw_slab_parts) is not
  // its grid -- so the dispatch is not run a second time in query mode on every replayed step (round 5 did)
  X3D_REQUIRE(!w->dw_slab || ((uintptr_t)w->dw_slab % 16) == 0, "pw_wgrad: dw_slab must be 16-byte aligned");
  if (w->dtype == X3D_F32) {
    if (a.stride == 1 && a.P >= 4 && x3d_env_int("X3D_PW_F32R", 1) != 0) {      // tile groups of <= 8, long double-buffered runs (pw_wgrad_f32r.h)
      PwWgradRArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.g = a.g; ra.yraw = a.yraw; ra.coef = a.coef; ra.fold = a.fold; ra.x = a.x; ra.xcoef = a.xcoef; ra.xgate = a.xgate; ra.xact = a.xact;
      ra.dw = a.dw; ra.slab = a.slab; ra.slab_parts = a.slab_parts; ra.N = a.N; ra.Cout = a.Cout; ra.Cin = a.Cin; ra.P = a.P;
      int rc = xpro ? wgrad_f32p_pick<true>(ra, st) : wgrad_f32p_pick<false>(ra, st);      // pipelined operand streams (pw_wgrad_f32p.h)
      if (rc < 0) rc = xpro ? wgrad_f32r_pick<true>(ra, st) : wgrad_f32r_pick<false>(ra, st);
      if (rc >= 0) return rc;
    }
    X3D_REQUIRE(!w->coef_fold, "pw_wgrad: coef_fold is not taken by the generic fp32 kernel (x3d_pw_coef_fold_supported() == 0)");
    if (x3d_parts_query) return X3D_OK;     // (query mode: the generic fp32 kernel has no slab form, the count stays 0)
    X3D_REQUIRE(!a.slab, "pw_wgrad: dw_slab given to a kernel without the slab form");
    return pw_wgrad_dispatch<float>(a, vec, xpro, st);
  }
  // bf16 storage: bf16 matrix cores; v2 = aligned fast path, v1 = generic (odd point counts / widths)
  if (w->dtype == X3D_F16) {
    const int rc = pw_wgrad_v2_dispatch<f16>(a, vec, xpro, st);
    if (rc < 0 && x3d_parts_query) return X3D_OK;       // (query mode: the generic kernel has no slab form)
    X3D_REQUIRE(rc >= 0 || !w->coef_fold, "pw_wgrad: coef_fold is not taken by the generic kernel (x3d_pw_coef_fold_supported() == 0)");
    return rc >= 0 ? rc : pw_wgrad_bf16_dispatch<f16>(a, vec, xpro, st);
  }
  const int rc = pw_wgrad_v2_dispatch<bf16>(a, vec, xpro, st);
  if (rc < 0 && x3d_parts_query) return X3D_OK;
  X3D_REQUIRE(rc >= 0 || !w->coef_fold, "pw_wgrad: coef_fold is not taken by the generic kernel (x3d_pw_coef_fold_supported() == 0)");
  return rc >= 0 ? rc : pw_wgrad_bf16_dispatch<bf16>(a, vec, xpro, st);
}

// number of partial slabs x3d_pw_wgrad writes when given dw_slab (0: no slab form behind this call): the whole dispatch runs
// in query mode -- the launcher reports its grid instead of launching
extern "C" int x3d_pw_wgrad_dw_parts(const x3d_pw_wgrad_args* w) {
  if (!w || !w->g || !w->x || x3d_describe.out) return 0;
  int parts = 0;
  int* saved = x3d_parts_query;
  x3d_parts_query = &parts;
  const int rc = x3d_pw_wgrad(w, nullptr);
  x3d_parts_query = saved;
  return rc == X3D_OK ? parts : 0;
}
