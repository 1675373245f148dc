// x3d_pw_fwd with the INFERENCE epilogue (EPI_BNADD, pw_gemm.h): y = act(s_o*acc + t_o [+ s_r*add + t_r]) -- the BatchNorm
// after the conv folded from the moving statistics and the residual Add + ReLU applied to the fp32 accumulators
// (reference model.py:300-303,368-371,381-392 at training=False).  The same kernels and dispatch as the training form
// (pw_fwd.hip); a translation unit of its own so the two sets of instantiations compile side by side.
#include "pw_gemm_wst.h"

template <typename H>
static int pw_fwd_bnadd_h16(PwGemmArgs& a, int vec, int ovec, bool pro, hipStream_t st) {
  const int shp_ = pw_wst_shape(a, vec, ovec);
  if (const int shp = ((shp_ == 5 && pro) || (shp_ >= 6 && pw_wst_shape_has_prologue(shp_) != pro)) ? 0 : shp_)
    return pro ? pw_wst_launch<H, PRO_AFFINE, EPI_BNADD>(a, shp, st) : pw_wst_launch<H, PRO_NONE, EPI_BNADD>(a, shp, st);
  if (pw_ws_applies(a, vec, ovec))
    return pro ? pw_ws_launch<H, PRO_AFFINE, EPI_BNADD>(a, st) : pw_ws_launch<H, PRO_NONE, EPI_BNADD>(a, st);
  return pro ? pw_bf16_launch_vec<H, PRO_AFFINE, EPI_BNADD>(a, vec, ovec, st)
             : pw_bf16_launch_vec<H, PRO_NONE, EPI_BNADD>(a, vec, ovec, st);
}

// called by x3d_pw_fwd (pw_fwd.hip) once the arguments are validated and `a` is filled
int pw_fwd_bnadd(PwGemmArgs& a, int dtype, int vec, int vec16, int ovec, bool pro, hipStream_t st) {
  if (dtype == X3D_F32)
    return pro ? pw_launch_vec<float, PRO_AFFINE, EPI_BNADD>(a, vec, st) : pw_launch_vec<float, PRO_NONE, EPI_BNADD>(a, vec, st);
  return dtype == X3D_F16 ? pw_fwd_bnadd_h16<f16>(a, vec16, ovec, pro, st) : pw_fwd_bnadd_h16<bf16>(a, vec16, ovec, pro, st);
}
