// x3d_pw_dgrad: pointwise convolution data gradient (see pw_gemm.h)
#include "pw_gemm_wst.h"
#include "pw_gemm_f32p.h"

template <typename T>
static int pw_dgrad_dispatch(PwGemmArgs& a, int epi, int vec, hipStream_t st) {
  {   // fp32 storage: weights resident in LDS, pipelined activation stream (pw_gemm_f32p.h) where the block fits
    int rc = -1;
    switch (epi) {
      case X3D_EPI_STORE: rc = f32p_try<PRO_BNBWD, X3D_EPI_STORE>(a, st); break;
      case X3D_EPI_ADD: rc = f32p_try<PRO_BNBWD, X3D_EPI_ADD>(a, st); break;
      // (X3D_EPI_ADD_STRIDED -- the first block of a stage, large planes, an integer division per output element -- stays with
      // pw_gemm_f32r.h: 54 -> 24 on 13x80x80 391 us there, 523 us in the pipelined kernel)
      case X3D_EPI_SWISH_BWD: rc = f32p_try<PRO_BNBWD, X3D_EPI_SWISH_BWD>(a, st); break;
    }
    if (rc >= 0) return rc;
    X3D_REQUIRE(!a.fold.sums, "pw_dgrad: coef_fold is not taken by the kernel behind this call (x3d_pw_coef_fold_supported() == 0)");
    switch (epi) {
      case X3D_EPI_STORE: rc = f32r_try<PRO_BNBWD, X3D_EPI_STORE>(a, vec, st); break;
      case X3D_EPI_ADD: rc = f32r_try<PRO_BNBWD, X3D_EPI_ADD>(a, vec, st); break;
      case X3D_EPI_ADD_STRIDED: rc = f32r_try<PRO_BNBWD, X3D_EPI_ADD_STRIDED>(a, vec, st); break;
      case X3D_EPI_SWISH_BWD: rc = f32r_try<PRO_BNBWD, X3D_EPI_SWISH_BWD>(a, vec, st); break;
    }
    if (rc >= 0) return rc;
  }
  switch (epi) {
    case X3D_EPI_STORE: return pw_launch_vec<T, PRO_BNBWD, X3D_EPI_STORE>(a, vec, st);
    case X3D_EPI_ADD: return pw_launch_vec<T, PRO_BNBWD, X3D_EPI_ADD>(a, vec, st);
    case X3D_EPI_ADD_STRIDED: return pw_launch_vec<T, PRO_BNBWD, X3D_EPI_ADD_STRIDED>(a, vec, st);
    case X3D_EPI_SWISH_BWD: return pw_launch_vec<T, PRO_BNBWD, X3D_EPI_SWISH_BWD>(a, vec, st);
  }
  x3d_set_error("pw_dgrad: unknown epilogue %d", epi);
  return X3D_ERR_INVALID;
}

template <typename H>
static int pw_dgrad_h16(PwGemmArgs& a, const x3d_pw_dgrad_args* d, int eb, int vec, hipStream_t st, int force_ovec = 0) {
  const int ovec = force_ovec ? force_ovec : pick_vec(eb, a.P, d->dx, d->epi == X3D_EPI_ADD ? d->add : nullptr, d->braw);
  // stage 5: weights stationary.  The stage-4 shapes stay with the resident-panel kernel here: with two staged tensors
  // the stationary kernel needs 132-146 VGPRs = one workgroup per CU (216 -> 96 dgrad: 58 -> 61 us)
  // (strided shortcut gradient: even image width only -- pairs of points never straddle a row)
  const int shp_ = (d->epi != X3D_EPI_ADD_STRIDED || (d->W % 2 == 0 && ((uintptr_t)d->add % 2) == 0)) ? pw_wst_shape(a, vec, ovec) : 0;
  const bool wst4 = x3d_env_int("X3D_PW_DGRAD_WST4", 0) == 1;   // experiment: the K = 96 -> M <= 224 stationary kernel for the stage-4 c-conv dgrad
  if (const int shp = (shp_ <= 2 || (shp_ == 4 && wst4)) ? shp_ : 0) {
    switch (d->epi) {
      case X3D_EPI_ADD_STRIDED: return pw_wst_launch<H, PRO_BNBWD, X3D_EPI_ADD_STRIDED>(a, shp, st);
      case X3D_EPI_STORE: return pw_wst_launch<H, PRO_BNBWD, X3D_EPI_STORE>(a, shp, st);
      case X3D_EPI_ADD: return pw_wst_launch<H, PRO_BNBWD, X3D_EPI_ADD>(a, shp, st);
      case X3D_EPI_SWISH_BWD: return pw_wst_launch<H, PRO_BNBWD, X3D_EPI_SWISH_BWD>(a, shp, st);
    }
  }
  // (coef_fold: the weights-stationary kernel above is the one data-gradient kernel that derives its table from the sums)
  X3D_REQUIRE(!d->coef_fold, "pw_dgrad: coef_fold is not taken by the kernel behind this call (x3d_pw_coef_fold_supported() == 0)");
  if (d->epi != X3D_EPI_ADD_STRIDED && pw_ws_applies(a, vec, ovec)) {   // deep, narrow layers (stage 5)
    switch (d->epi) {
      case X3D_EPI_STORE: return pw_ws_launch<H, PRO_BNBWD, X3D_EPI_STORE>(a, st);
      case X3D_EPI_ADD: return pw_ws_launch<H, PRO_BNBWD, X3D_EPI_ADD>(a, st);
      case X3D_EPI_SWISH_BWD: return pw_ws_launch<H, PRO_BNBWD, X3D_EPI_SWISH_BWD>(a, st);
    }
  }
  switch (d->epi) {  // bf16 storage: bf16 matrix cores
    case X3D_EPI_STORE: return pw_bf16_launch_vec<H, PRO_BNBWD, X3D_EPI_STORE>(a, vec, ovec, st);
    case X3D_EPI_ADD: return pw_bf16_launch_vec<H, PRO_BNBWD, X3D_EPI_ADD>(a, vec, ovec, st);
    case X3D_EPI_ADD_STRIDED: return pw_bf16_launch_vec<H, PRO_BNBWD, X3D_EPI_ADD_STRIDED>(a, vec, ovec, st);
    case X3D_EPI_SWISH_BWD: return pw_bf16_launch_vec<H, PRO_BNBWD, X3D_EPI_SWISH_BWD>(a, vec, ovec, st);
  }
  x3d_set_error("pw_dgrad: unknown epilogue %d", d->epi);
  return X3D_ERR_INVALID;
}

extern "C" int x3d_pw_dgrad(const x3d_pw_dgrad_args* d, void* stream) {
  X3D_REQUIRE(d && d->g && d->w && d->dx, "pw_dgrad: null pointer");
  X3D_REQUIRE((d->coef || d->coef_fold) && d->yraw, "pw_dgrad: coef (or coef_fold) / yraw required (every conv on the path feeds a BN)");
  X3D_REQUIRE(x3d_describe.out || bn_bwd_fold_ok(d->coef_fold), "pw_dgrad: incomplete coef_fold");
  X3D_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->T > 0 && d->H > 0 && d->W > 0,
              "pw_dgrad: bad extents");
  X3D_REQUIRE(x3d_dtype_ok(d->dtype), "pw_dgrad: bad dtype");
  if (d->epi == X3D_EPI_ADD || d->epi == X3D_EPI_ADD_STRIDED)
    X3D_REQUIRE(d->add, "pw_dgrad: epilogue needs `add`");
  if (d->epi == X3D_EPI_SWISH_BWD)
    X3D_REQUIRE(d->braw && d->b_scale_shift && d->nc_sums, "pw_dgrad: SWISH_BWD needs braw/b_scale_shift/nc_sums");
  X3D_REQUIRE(((uintptr_t)d->w_panel % 16) == 0, "pw_dgrad: w_panel must be 16-byte aligned");
  PwGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.x = d->g; a.x2 = d->yraw; a.coef = d->coef; a.fold = bn_bwd_fold_arg(d->coef_fold);
  a.w = d->w; a.wsk = d->Cin; a.wsm = 1;  // (k = co, m = ci) -> w[co*Cin + ci]
  a.N = d->N; a.K = d->Cout; a.M = d->Cin;
  a.stride = 1;
  a.P = a.Pin = (long long)d->T * d->H * d->W;
  X3D_REQUIRE(a.Pin < (1ll << 31) && a.P < (1ll << 31), "pw_dgrad: more than 2^31 points per sample");   // 32-bit point indices in the kernels
  a.y = d->dx; a.add = d->add; a.braw = d->braw; a.b_ss = d->b_scale_shift; a.egate = d->gate;
  a.nc_sums = d->nc_sums; a.eH = d->H; a.eW = d->W;
  a.wp = d->w_panel; a.wp_rows = (d->Cin + 31) & ~31;
  const int eb = d->dtype == X3D_F32 ? 4 : 2;
  const int vec = pick_vec(eb, a.P, d->g, d->yraw);
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == X3D_F32) return pw_dgrad_dispatch<float>(a, d->epi, vec, st);
  const bool al16 = (((uintptr_t)d->g | (uintptr_t)d->yraw | (uintptr_t)d->dx | (uintptr_t)d->braw |
                      (d->epi == X3D_EPI_ADD ? (uintptr_t)d->add : 0)) % 16) == 0;
  if (al16 && pw_ragged_rows(a.P, eb))   // P % 8 != 0: the vector form with ragged row ends (pw_gemm.h)
    return d->dtype == X3D_F16 ? pw_dgrad_h16<f16>(a, d, eb, 8, st, 8) : pw_dgrad_h16<bf16>(a, d, eb, 8, st, 8);
  return d->dtype == X3D_F16 ? pw_dgrad_h16<f16>(a, d, eb, vec, st) : pw_dgrad_h16<bf16>(a, d, eb, vec, st);
}

// does the kernel behind a call derive its coefficient table from the BatchNorm-backward sums (include/x3d_hip.h coef_fold)?
// Asked of the dispatch itself: the name of the instantiation the call would launch.
extern "C" int x3d_pw_coef_fold_supported(const x3d_pw_dgrad_args* dgrad, const x3d_pw_wgrad_args* wgrad, const x3d_pw_bwd_args* bwd) {
  if ((dgrad != nullptr) + (wgrad != nullptr) + (bwd != nullptr) != 1) return 0;
  char name[160];
  int rc;
  if (dgrad) { x3d_pw_dgrad_args t = *dgrad; t.coef_fold = nullptr; if (!t.coef) t.coef = (const float*)16; if (!t.nc_sums) t.nc_sums = (double*)16; rc = x3d_pw_kernel_name(nullptr, &t, nullptr, nullptr, name, sizeof(name)); }
  else if (wgrad) { x3d_pw_wgrad_args t = *wgrad; t.coef_fold = nullptr; if (!t.coef) t.coef = (const float*)16; rc = x3d_pw_kernel_name(nullptr, nullptr, &t, nullptr, name, sizeof(name)); }
  else {   // (a plan asks before it has resolved its accumulators: the dry-run dispatch only looks at what is NULL)
    x3d_pw_bwd_args t = *bwd; t.coef_fold = nullptr; if (!t.coef) t.coef = (const float*)16; if (!t.nc_sums) t.nc_sums = (double*)16;
    rc = x3d_pw_kernel_name(nullptr, nullptr, nullptr, &t, name, sizeof(name));
  }
  if (rc != X3D_OK) return 0;
  static const char* const ok[] = {"pw_bwd_wst_kernel<", "pw_bwd_wsta_kernel<", "pw_wgrad_bf16_v2_kernel<"};
  for (const char* p : ok) if (strncmp(name, p, strlen(p)) == 0) return 1;
  if (strncmp(name, "pw_gemm_wst_kernel<", 19) == 0) return dgrad != nullptr;
  if (strncmp(name, "pw_f32p_kernel<", 15) == 0) return dgrad != nullptr;          // fp32 storage: the pipelined data-gradient kernel ...
  if (strncmp(name, "pw_wgrad_f32r_kernel<", 21) == 0 || strncmp(name, "pw_wgrad_f32p_kernel<", 21) == 0) return wgrad != nullptr;    // ... and the tile-group weight-gradient kernel
  return 0;
}
