// x3d_dw3d_fwd: channelwise 3x3x3 convolution forward (design notes in dw_common.h)
#include "dw_common.h"


// ================================================================================================
// forward.  NSV = staging vectors per thread held in registers for the prefetch (0: direct staging)
// ================================================================================================
// CV > 0: the staging vector width (CV elements) and SW-aligned strips are guaranteed by the host, so every
// width / alignment decision is compile-time (the runtime-width path costs ~300 scalar instructions per plane)
// DI (stride 2, strips of four, 16-byte staging vectors of 16-bit elements, no left pad): the LDS plane is DE-INTERLEAVED -- every
// row holds its even columns, then its odd columns (half pitch LPh).  With the plain layout a thread's window starts 8 floats
// after its neighbour's and its staging vector lands 8 floats after its neighbour's: lanes 32 bytes apart meet on the same
// banks (SQ_LDS_BANK_CONFLICT = 74 % of SQ_LDS_IDX_ACTIVE, the LDS pipe busy 68 % of the launch).  De-interleaved, a staging
// vector is two 16-byte writes and a window two 16-byte reads (+ one float) at a lane stride of 16 bytes: conflict-free.
template <typename T, int S, int SW, int NSV, int CV, bool DI>
__device__ __forceinline__ void dw3d_fwd_body(const DwFwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwGeom& g = a.g;
  static_assert(!DI || (S == 2 && SW == 4 && CV == 8 && NSV > 0 && sizeof(T) == 2), "de-interleaved planes: stride 2, strips of 4, 8-element vectors");
  const int LPh = (g.Wo + 4) & ~3;      // DI: floats per half row (even columns 0 .. Wo: the last one is the right halo)
  constexpr int WIN = (SW - 1) * S + 3;
  constexpr int NS = NSV > 0 ? NSV : 1;
  constexpr bool RAG = CV < 0;            // ragged plane: flat staging in vectors of -CV elements (FlatMap, dw_common.h)
  constexpr int RV = RAG ? -CV : 1;
  const int plane_sz = g.RIN * (DI ? 2 * LPh : g.LP);
  float* scratch = lds + plane_sz;
#ifdef X3D_EXPERIMENTS
  const bool x_nomath = a.exp & 1, x_noload = a.exp & 2;
#else
  constexpr bool x_nomath = false, x_noload = false;
#endif

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

  const long long iplane = (long long)g.H * g.W, oplane = (long long)g.Ho * g.Wo;
  const T* xin = (const T*)a.x + ((long long)n * g.C + c) * g.T * iplane;
  T* yout = (T*)a.y + ((long long)n * g.C + c) * g.T * oplane;
  const int row0 = h0 * S - g.ph;
  const int vec = CV > 0 ? CV : (RAG ? RV : g.vec);

  typename DwSel<RAG, FlatMap<NS>, StageMap<NS>>::type map;
  Raw raw[NS];
  if constexpr (NSV > 0) {
    map.build(g.RIN, g.LP, row0, g.H, g.W, g.pw, vec);
#pragma unroll
    for (int i = 0; i < NS; i++) if (map.goff[i] >= 0 && !x_noload) raw_load<T>(raw[i], xin + map.goff[i], vec);
    if constexpr (DI) {   // loff = row * LP + column (pw == 0)  ->  row * 2 LPh + column / 2
#pragma unroll
      for (int i = 0; i < NS; i++) {
        const int lr = map.loff[i] / g.LP, c0 = map.loff[i] - lr * g.LP;
        map.loff[i] = lr * 2 * LPh + (c0 >> 1);
      }
    }
  }

  v2f acc01[SW], acc2p[(SW + 1) / 2];   // (out[t-1], out[t]) per output, out[t+1] as pairs over outputs
#pragma unroll
  for (int i = 0; i < SW; i++) acc01[i] = (v2f){0.f, 0.f};
#pragma unroll
  for (int j = 0; j < (SW + 1) / 2; j++) acc2p[j] = (v2f){0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  float fin[SW];   // finished output plane waiting for its (deferred) store
#pragma unroll
  for (int i = 0; i < SW; i++) fin[i] = 0.f;

  // every strip is full and SW-aligned when Wo % SW == 0: one 8/16-byte store per thread and plane instead
  // of SW two-byte stores (the scalar stores, not HBM, were the limiter of the stride-1 layers)
  // (ragged planes: every strip that is whole, at whatever alignment; the row's last strip stores its elements one by one)
  const bool vstore = RAG ? (SW > 1 && wo0 + SW <= g.Wo)
                          : ((CV > 0 && SW > 1) || ((SW > 1) && (g.Wo % SW == 0) && (((uintptr_t)a.y) % (SW * sizeof(T)) == 0)));
  auto store_plane = [&](int t, const float (&v)[SW]) {
    if (!active || x_noload) return;
    T* dst = yout + t * oplane + (long long)ho * g.Wo + wo0;
    if (vstore) {
      VecIO<T, SW>::store(dst, v);
#pragma unroll
      for (int i = 0; i < SW; i++) {
        s1 += v[i];
        s2 += v[i] * v[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < SW; i++) {
        if (wo0 + i < g.Wo) {
          dst[i] = from_f<T>(v[i]);
          s1 += v[i];
          s2 += v[i] * v[i];
        }
      }
    }
  };

  for (int t = 0; t < g.T; ++t) {
    __syncthreads();  // zero-fill / previous plane's readers done
    if constexpr (NSV > 0) {
#pragma unroll
      for (int i = 0; i < NS; i++) {
        if (map.goff[i] >= 0) {
          float* d = lds + map.loff[i];
          if constexpr (RAG) flat_commit<T, RV>(d, map.wrap[i], g.LP - g.W, raw[i], xf);
          else if constexpr (DI) {
            f32x4 ev, od;
#pragma unroll
            for (int e = 0; e < 4; e++) { ev[e] = xf(raw_get<T>(raw[i], 2 * e)); od[e] = xf(raw_get<T>(raw[i], 2 * e + 1)); }
            *(f32x4*)d = ev;
            *(f32x4*)(d + LPh) = od;
          } else {
#pragma unroll
            for (int e = 0; e < MaxVec<T>::v; e++) if (e < vec) d[e] = xf(raw_get<T>(raw[i], e));
          }
        }
      }
    } else {
      stage_direct<T>(xin + t * iplane, lds, g.RIN, g.LP, row0, g.H, g.W, g.pw, vec, xf);
    }
    __syncthreads();
    if constexpr (NSV > 0) {  // next plane's loads fly while this one is consumed
      if (t + 1 < g.T) {
#pragma unroll
        for (int i = 0; i < NS; i++) if (map.goff[i] >= 0 && !x_noload) raw_load<T>(raw[i], xin + (t + 1) * iplane + map.goff[i], vec);
      }
    }
    if (t >= 2) store_plane(t - 2, fin);
    if (active && !x_nomath) {
#pragma unroll
      for (int kh = 0; kh < 3; kh++) {
        float win[WIN];
        if constexpr (DI) {
          const float* row = lds + (r * S + kh) * 2 * LPh + wo0;
          const f32x4 ev = *(const f32x4*)row, od = *(const f32x4*)(row + LPh);
#pragma unroll
          for (int e = 0; e < 4; e++) { win[2 * e] = ev[e]; win[2 * e + 1] = od[e]; }
          win[8] = row[4];
        } else {
        const float* row = lds + (r * S + kh) * g.LP + wo0 * S;   // wo0 * S is a multiple of SW * S floats
        lds_window<WIN, (SW * S >= 4 ? 4 : SW * S)>(row, win);
        }
        // out[t-1] sees this plane through kt = 2, out[t] through kt = 1, out[t+1] through kt = 0
        dw_taps_row<S, SW, WIN>(win, w21[kh], w0[kh], acc01, acc2p);
      }
    }
    // acc0 now holds output plane t-1 complete.  Its store is DEFERRED to the start of the next iteration's
    // arithmetic: vmcnt retires in order, so a store issued here would sit in front of the wait for the
    // prefetched plane at the top of the next iteration and expose its full write latency every plane.
    dw_rotate<SW>(fin, acc01, acc2p);
  }
  if (g.T >= 2) store_plane(g.T - 2, fin);
  float last[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) last[i] = acc01[i].x;
  store_plane(g.T - 1, last);

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

template <typename T, int S, int SW, int NSV, int CV>
__global__ __launch_bounds__(256) void dw3d_fwd_kernel(const DwFwdArgs a) { dw3d_fwd_body<T, S, SW, NSV, CV, false>(a); }
template <typename T, int NSV>
__global__ __launch_bounds__(256) void dw3d_fwd_di_kernel(const DwFwdArgs a) { dw3d_fwd_body<T, 2, 4, NSV, 8, true>(a); }

template <typename T, int S, int SW, int CV>
static void dw_fwd_launch_cv(const DwFwdArgs& a, int nsv, unsigned grid, int bd, size_t lds, hipStream_t st) {
  if (x3d_describe.out) {
    const bool gen = nsv > 4;
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_fwd_kernel<%s, %d, %d, %d, %d>", TypeName<T>::v,
             S, SW, gen ? 0 : (nsv <= 2 ? 2 : 4), gen ? 0 : CV);
    return;
  }
  if (nsv <= 2) hipLaunchKernelGGL((dw3d_fwd_kernel<T, S, SW, 2, CV>), dim3(grid), dim3(bd), lds, st, a);
  else if (nsv <= 4) hipLaunchKernelGGL((dw3d_fwd_kernel<T, S, SW, 4, CV>), dim3(grid), dim3(bd), lds, st, a);
  else hipLaunchKernelGGL((dw3d_fwd_kernel<T, S, SW, 0, 0>), dim3(grid), dim3(bd), lds, st, a);
}
template <typename T, int S, int SW>
static void dw_fwd_launch_nsv(const DwFwdArgs& a, int nsv, int cv, unsigned grid, int bd, size_t lds, hipStream_t st) {
  switch (cv) {
    case -8: if constexpr (sizeof(T) == 2) { dw_fwd_launch_cv<T, S, SW, -8>(a, nsv, grid, bd, lds, st); break; }
    case -4: dw_fwd_launch_cv<T, S, SW, -4>(a, nsv, grid, bd, lds, st); break;
    case 8: if constexpr (sizeof(T) == 2) { dw_fwd_launch_cv<T, S, SW, 8>(a, nsv, grid, bd, lds, st); break; }
    case 4: dw_fwd_launch_cv<T, S, SW, 4>(a, nsv, grid, bd, lds, st); break;
    case 2: dw_fwd_launch_cv<T, S, SW, 2>(a, nsv, grid, bd, lds, st); break;
    case 1: dw_fwd_launch_cv<T, S, SW, 1>(a, nsv, grid, bd, lds, st); break;
    default: dw_fwd_launch_cv<T, S, SW, 0>(a, nsv, grid, bd, lds, st); break;
  }
}

template <typename T, int S>
static int dw_fwd_launch(const x3d_dw3d_fwd_args* f, hipStream_t st) {
  DwFwdArgs a;
  a.x = f->x; a.w = f->w; a.y = f->y; a.ss = f->in_scale_shift; a.act = f->in_act;
  a.stats = f->stats; a.pool = f->pool;
  memset(&a.bn, 0, sizeof(a.bn));
  if (f->in_bn) a.bn = *f->in_bn;
  a.exp = x3d_env_int("X3D_DW_FWD_EXP", 0);   // result-changing timing hooks: -DX3D_EXPERIMENTS builds only
  const int Wo = ceil_div(f->W, S);
  const int SW = dw_pick_sw(Wo);
  int bd; size_t ldsf;
  if (dw_geom(a.g, f->N, f->C, f->T, f->H, f->W, S, SW, sizeof(T), f->x, nullptr, nullptr, &bd, &ldsf)) {
    x3d_set_error("dw3d_fwd: row of %d outputs does not fit one workgroup", Wo);
    return X3D_ERR_INVALID;
  }
  const size_t lds = (ldsf + 2 * 4 + 8) * sizeof(float);
  X3D_REQUIRE(lds <= 64 * 1024, "dw3d_fwd: tile needs %zu B of LDS", lds);
  const long long grid = (long long)f->N * f->C * a.g.ntile_h;
  X3D_REQUIRE(grid < (1ll << 31), "dw3d_fwd: grid too large");
  int nsv = dw_nsv(a.g.RIN, a.g.W, a.g.vec, bd);
  // compile-time staging width when every output strip is whole and SW-aligned (always true for SW == 1)
  const bool strips_ok = (a.g.Wo % SW == 0) && (((uintptr_t)f->y) % (SW * sizeof(T)) == 0);
  int cv = strips_ok ? a.g.vec : 0;
  // small planes: deep-prefetch variant (dw_pd.hip) when one staging vector per thread covers the tile
  const int pd = dw_pick_pd(SW);
  if (dw_fwd_mx_launch(a, f->dtype, S, st)) {   // 14x14 stride-1 planes, 16-bit storage: the tap sums on the matrix cores (dw_mx.hip)
    if (x3d_describe.out) return X3D_OK;
    X3D_LAUNCH_CHECK("dw3d_fwd");
    return X3D_OK;
  }
  if (cv > 0 && dw_fwd_pk_launch(a, f->dtype, S, SW, st)) {   // 10..18-wide and 7x7 stride-1 planes: packed kernel (dw_pk.hip)
    if (x3d_describe.out) return X3D_OK;
    X3D_LAUNCH_CHECK("dw3d_fwd");
    return X3D_OK;
  }
  if (pd > 1 && cv > 0 && nsv <= 1 && (long long)f->T * f->H * f->W * (long long)sizeof(T) < (1ll << 30)) {
    if (dw_fwd_pd_launch(a, f->dtype, S, SW, cv, pd, (unsigned)grid, bd, lds, st)) {
      if (x3d_describe.out) return X3D_OK;
      X3D_LAUNCH_CHECK("dw3d_fwd");
      return X3D_OK;
    }
  }
  // ragged rows (39, 78, 91 ... wide, or a misaligned tensor): flat staging with unaligned 16 / 8-byte vectors instead
  // of the run-time-width path.  X3D_DW_FLAT=0: A/B hook.
  const int flat_env = x3d_env_int("X3D_DW_FLAT", 1);
  // (also when the strips are whole but the rows only admit 2 / 4-byte vectors: 39 -> 20, 23 -> 12 at stride 2)
  const bool narrow = cv > 0 && cv * sizeof(T) < 8;
  if ((cv == 0 || narrow) && flat_env != 0) {
    const int rv = dw_flat_vec(sizeof(T), a.g.W);
    if (rv > 0 && dw_nsv_flat(a.g.RIN, a.g.W, rv, bd) <= 4) { cv = -rv; nsv = dw_nsv_flat(a.g.RIN, a.g.W, rv, bd); }
  }
  // stride 2, strips of four, whole 16-byte vectors, no left pad: the de-interleaved LDS plane.  X3D_DW_FWD_DI=0: A/B hook
  if constexpr (S == 2 && sizeof(T) == 2) {
    if (SW == 4 && cv == 8 && a.g.pw == 0 && nsv <= 4 && x3d_env_int("X3D_DW_FWD_DI", 1) != 0) {
      const int LPh = (a.g.Wo + 4) & ~3;
      const size_t lds_di = ((size_t)a.g.RIN * 2 * LPh + 2 * 4 + 8) * sizeof(float);
      if (lds_di <= 64 * 1024) {
        if (x3d_describe.out) {
          snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_fwd_di_kernel<%s, %d>", TypeName<T>::v, nsv <= 2 ? 2 : 4);
          return X3D_OK;
        }
        if (nsv <= 2) hipLaunchKernelGGL((dw3d_fwd_di_kernel<T, 2>), dim3((unsigned)grid), dim3(bd), lds_di, st, a);
        else hipLaunchKernelGGL((dw3d_fwd_di_kernel<T, 4>), dim3((unsigned)grid), dim3(bd), lds_di, st, a);
        X3D_LAUNCH_CHECK("dw3d_fwd");
        return X3D_OK;
      }
    }
  }
  switch (SW) {
    case 4: dw_fwd_launch_nsv<T, S, 4>(a, nsv, cv, (unsigned)grid, bd, lds, st); break;
    case 2: dw_fwd_launch_nsv<T, S, 2>(a, nsv, cv, (unsigned)grid, bd, lds, st); break;
    default: dw_fwd_launch_nsv<T, S, 1>(a, nsv, cv, (unsigned)grid, bd, lds, st); break;
  }
  if (x3d_describe.out) return X3D_OK;
  X3D_LAUNCH_CHECK("dw3d_fwd");
  return X3D_OK;
}

extern "C" int x3d_dw3d_kernel_name(const x3d_dw3d_fwd_args* fwd, const x3d_dw3d_bwd_args* bwd, char* out, int cap) {
  X3D_REQUIRE(out && cap > 0 && ((fwd != nullptr) != (bwd != nullptr)), "dw3d_kernel_name: pass exactly one of fwd / bwd");
  out[0] = 0;
  x3d_describe = {out, cap};
  const int rc = fwd ? x3d_dw3d_fwd(fwd, nullptr) : x3d_dw3d_bwd(bwd, nullptr);
  x3d_describe = {nullptr, 0};
  return rc;
}

extern "C" int x3d_dw3d_fwd(const x3d_dw3d_fwd_args* f, void* stream) {
  X3D_REQUIRE(f && f->x && f->w && f->y, "dw3d_fwd: null pointer");
  X3D_REQUIRE(f->stride == 1 || f->stride == 2, "dw3d_fwd: stride must be 1 or 2");
  X3D_REQUIRE(f->N > 0 && f->C > 0 && f->T > 0 && f->H > 0 && f->W > 0, "dw3d_fwd: bad extents");
  X3D_REQUIRE(x3d_dtype_ok(f->dtype), "dw3d_fwd: bad dtype");
  X3D_REQUIRE(f->in_act == X3D_ACT_NONE || f->in_act == X3D_ACT_RELU, "dw3d_fwd: prologue act must be none/relu");
  X3D_REQUIRE(!f->in_bn || bn_fold_valid(f->in_bn), "dw3d_fwd: incomplete x3d_bn_fold");
  hipStream_t st = (hipStream_t)stream;
  if (f->dtype == X3D_F32)
    return f->stride == 1 ? dw_fwd_launch<float, 1>(f, st) : dw_fwd_launch<float, 2>(f, st);
  if (f->dtype == X3D_F16)
    return f->stride == 1 ? dw_fwd_launch<f16, 1>(f, st) : dw_fwd_launch<f16, 2>(f, st);
  return f->stride == 1 ? dw_fwd_launch<bf16, 1>(f, st) : dw_fwd_launch<bf16, 2>(f, st);
}

