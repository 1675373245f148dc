// x3d_pw_fwd: pointwise convolution forward (see pw_gemm.h)
#include "pw_gemm_wst.h"
#include "pw_gemm_f32p.h"

template <typename H>
static int pw_fwd_h16(PwGemmArgs& a, int vec, int ovec, bool pro, hipStream_t st) {
  const int shp_ = pw_wst_shape(a, vec, ovec);       // stage 4 / 5: weights stationary in registers
  if (const int shp = ((shp_ == 5 && pro) || (shp_ >= 6 && pw_wst_shape_has_prologue(shp_) != pro)) ? 0 : shp_)  // (shape 5 with a prologue: 129 VGPRs, one workgroup per CU)
    return pro ? pw_wst_launch<H, PRO_AFFINE, EPI_STATS>(a, shp, st) : pw_wst_launch<H, PRO_NONE, EPI_STATS>(a, shp, st);
  if (pw_ws_applies(a, vec, ovec))   // deep, narrow layers (stage 5): weights streamed, 32-point tiles
    return pro ? pw_ws_launch<H, PRO_AFFINE, EPI_STATS>(a, st) : pw_ws_launch<H, PRO_NONE, EPI_STATS>(a, st);
  return pro ? pw_bf16_launch_vec<H, PRO_AFFINE, EPI_STATS>(a, vec, ovec, st)
             : pw_bf16_launch_vec<H, PRO_NONE, EPI_STATS>(a, vec, ovec, st);
}

int pw_fwd_bnadd(PwGemmArgs& a, int dtype, int vec, int vec16, int ovec, bool pro, hipStream_t st);   // pw_fwd_infer.hip
int pw_fwd_tail(PwGemmArgs& a, int dtype, int vec, int ovec, hipStream_t st);                          // pw_fwd_tail.hip
int pw_fwd_affst(PwGemmArgs& a, int dtype, int vec, int ovec, hipStream_t st);

// the folded residual tail (in_add / in_store) or, without in_add, the folded BatchNorm + ReLU pass of the stem: 16-bit
// storage, stride 1, BN + ReLU prologue without a gate, training epilogue
extern "C" int x3d_pw_fwd_tail_supported(const x3d_pw_fwd_args* f) {
  if (!f || !f->in_store || !f->in_scale_shift || (!f->in_add && f->in_add_scale_shift)) return 0;
  if (f->stride != 1 || f->in_gate || f->in_act != X3D_ACT_RELU || f->out_scale_shift) return 0;
  if (f->dtype == X3D_F32) {      // fp32 storage: the resident-weights kernel carries the fold (pw_gemm_f32r.h) where it covers the layer
    PwGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.K = f->Cin; a.M = f->Cout; a.stride = 1; a.P = (long long)f->T * f->H * f->W;
    int MT, NT;
    return (f32r_enabled() && a.P >= 4 && f32r_shape(a, 2, &MT, &NT)) ? 1 : 0;
  }
  return 1;
}

extern "C" int x3d_pw_fwd(const x3d_pw_fwd_args* f, void* stream) {
  X3D_REQUIRE(f && f->x && f->w && f->y, "pw_fwd: null pointer");
  X3D_REQUIRE(f->stride == 1 || f->stride == 2, "pw_fwd: stride must be 1 or 2");
  X3D_REQUIRE(f->N > 0 && f->Cin > 0 && f->Cout > 0 && f->T > 0 && f->H > 0 && f->W > 0,
              "pw_fwd: bad extents");
  X3D_REQUIRE(x3d_dtype_ok(f->dtype), "pw_fwd: bad dtype");
  X3D_REQUIRE(!(f->stride > 1 && f->in_scale_shift), "pw_fwd: strided input takes no prologue");
  X3D_REQUIRE(((uintptr_t)f->w_panel % 16) == 0, "pw_fwd: w_panel must be 16-byte aligned");
  PwGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.x = f->x; a.coef = f->in_scale_shift; a.gate = f->in_gate; a.act = f->in_act;
  a.w = f->w; a.wsk = 1; a.wsm = f->Cin;  // (k = ci, m = co) -> w[co*Cin + ci]
  a.N = f->N; a.K = f->Cin; a.M = f->Cout;
  a.stride = f->stride; a.H = f->H; a.W = f->W;
  a.Ho = ceil_div(f->H, f->stride); a.Wo = ceil_div(f->W, f->stride);
  a.Pin = (long long)f->T * f->H * f->W;
  a.P = (long long)f->T * a.Ho * a.Wo;
  X3D_REQUIRE(a.Pin < (1ll << 31) && a.P < (1ll << 31), "pw_fwd: more than 2^31 points per sample");   // 32-bit point indices in the kernels
  a.y = f->y; a.stats = f->stats;
  a.wp = f->w_panel; a.wp_rows = (f->Cout + 31) & ~31;
  hipStream_t st = (hipStream_t)stream;
  const int eb = f->dtype == X3D_F32 ? 4 : 2;
  const int vec = pick_vec(eb, a.P, f->x);
  const bool pro = f->in_scale_shift != nullptr || f->in_gate != nullptr || f->in_act != X3D_ACT_NONE;
  X3D_REQUIRE(!pro || f->in_scale_shift, "pw_fwd: gate/activation prologue needs in_scale_shift");
  if (f->in_add || f->in_store || f->in_add_scale_shift) {      // residual tail of the block below folded into the prologue
    X3D_REQUIRE(x3d_pw_fwd_tail_supported(f), "pw_fwd: in_add / in_store (folded residual tail) need 16-bit storage, stride 1, "
                                              "in_scale_shift + ReLU, no gate (x3d_pw_fwd_tail_supported)");
    a.x2 = f->in_add; a.coef2 = f->in_add_scale_shift; a.ystore = f->in_store;
    const int vt = pick_vec(eb, a.P, f->x, f->in_add ? f->in_add : f->x, f->in_store);
    int ot = pick_vec(eb, a.P, f->y);
    int v16 = vt;
    if (((uintptr_t)f->x % 16) == 0 && ((uintptr_t)f->in_add % 16) == 0 && ((uintptr_t)f->in_store % 16) == 0 &&
        ((uintptr_t)f->y % 16) == 0 && pw_ragged_rows(a.P, eb)) v16 = ot = 8;
    if (f->dtype == X3D_F32) {
      // (the stem's BatchNorm fold -- one launch, 24 -> 54 on 13x80x80, one chunk a tile -- stays with pw_gemm_f32r.h: 221 us there, 265 us
      // in the pipelined kernel, profiles/r06_f32p_layers.txt)
      int rc = f->in_add ? f32p_try<PRO_TAIL, EPI_STATS>(a, st) : -1;
      if (rc < 0) rc = f->in_add ? f32r_try<PRO_TAIL, EPI_STATS>(a, 4, st) : f32r_try<PRO_AFFST, EPI_STATS>(a, 4, st);
      X3D_REQUIRE(rc >= 0, "pw_fwd: folded tail in fp32 storage: layer not covered by the resident-weights kernel");
      return rc;
    }
    return f->in_add ? pw_fwd_tail(a, f->dtype, v16, ot, st) : pw_fwd_affst(a, f->dtype, v16, ot, st);
  }
  if (f->out_scale_shift) {      // inference epilogue: folded BN + residual Add + activation on the accumulators
    X3D_REQUIRE(!f->stats, "pw_fwd: the inference epilogue (out_scale_shift) takes no statistics");
    X3D_REQUIRE(f->stride == 1, "pw_fwd: the inference epilogue is for stride 1 (the strided shortcut stays raw)");
    X3D_REQUIRE(f->out_act == X3D_ACT_NONE || f->out_act == X3D_ACT_RELU, "pw_fwd: out_act must be none or ReLU");
    X3D_REQUIRE(!f->out_add_scale_shift || f->out_add, "pw_fwd: out_add_scale_shift without out_add");
    a.e_ss = f->out_scale_shift; a.e_ass = f->out_add_scale_shift; a.add = f->out_add; a.eact = f->out_act;
    int ovec_ = pick_vec(eb, a.P, f->y);
    if (f->out_add) ovec_ = ovec_ < pick_vec(eb, a.P, f->out_add) ? ovec_ : pick_vec(eb, a.P, f->out_add);
    int v16 = vec;
    if (f->dtype != X3D_F32 && ((uintptr_t)f->x % 16) == 0 && ((uintptr_t)f->y % 16) == 0 &&
        (!f->out_add || ((uintptr_t)f->out_add % 16) == 0) && pw_ragged_rows(a.P, eb)) v16 = ovec_ = 8;
    return pw_fwd_bnadd(a, f->dtype, vec, v16, ovec_, pro, st);
  }
  X3D_REQUIRE(!f->out_add && !f->out_add_scale_shift, "pw_fwd: out_add needs out_scale_shift");
  if (f->dtype == X3D_F32) {
    // weights resident in LDS, activations double-buffered (pw_gemm_f32r.h); the strided shortcut keeps the per-tile kernel
    int rc = pro ? f32p_try<PRO_AFFINE, EPI_STATS>(a, st) : f32p_try<PRO_NONE, EPI_STATS>(a, st);
    if (rc < 0) rc = pro ? f32r_try<PRO_AFFINE, EPI_STATS>(a, vec, st) : f32r_try<PRO_NONE, EPI_STATS>(a, vec, st);
    if (rc >= 0) return rc;
    return pro ? pw_launch_vec<float, PRO_AFFINE, EPI_STATS>(a, vec, st)
               : pw_launch_vec<float, PRO_NONE, EPI_STATS>(a, vec, st);
  }
  // 16-bit storage: bf16 / f16 matrix cores (fp32 accumulate)
  int ovec = pick_vec(eb, a.P, f->y);
  int vec16 = vec;
  if (((uintptr_t)f->x % 16) == 0 && ((uintptr_t)f->y % 16) == 0 && pw_ragged_rows(a.P, eb)) vec16 = ovec = 8;   // (stride 2: the gather groups)
  if (vec16 != vec) return f->dtype == X3D_F16 ? pw_fwd_h16<f16>(a, vec16, ovec, pro, st) : pw_fwd_h16<bf16>(a, vec16, ovec, pro, st);
  return f->dtype == X3D_F16 ? pw_fwd_h16<f16>(a, vec, ovec, pro, st) : pw_fwd_h16<bf16>(a, vec, ovec, pro, st);
}


// dry-run dispatch of the four pointwise entry points (include/x3d_hip.h): the launchers stop at X3D_DESCRIBE
extern "C" int x3d_pw_kernel_name(const x3d_pw_fwd_args* fwd, const x3d_pw_dgrad_args* dgrad, const x3d_pw_wgrad_args* wgrad,
                                  const x3d_pw_bwd_args* bwd, char* out, int cap) {
  X3D_REQUIRE(out && cap > 0, "pw_kernel_name: no output buffer");
  X3D_REQUIRE((fwd != nullptr) + (dgrad != nullptr) + (wgrad != nullptr) + (bwd != nullptr) == 1,
              "pw_kernel_name: exactly one argument struct");
  out[0] = 0;
  x3d_describe = {out, cap};
  const int rc = fwd ? x3d_pw_fwd(fwd, nullptr) : dgrad ? x3d_pw_dgrad(dgrad, nullptr) : wgrad ? x3d_pw_wgrad(wgrad, nullptr)
                                                                                       : x3d_pw_bwd(bwd, nullptr);
  x3d_describe = {nullptr, 0};
  if (rc == X3D_OK && out[0] == 0) { x3d_set_error("pw_kernel_name: the dispatch launched nothing"); return X3D_ERR_INVALID; }
  return rc;
}
