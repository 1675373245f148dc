// x3d_pw_fwd with the residual tail of the block below folded into the prologue (PRO_TAIL, pw_gemm.h): the `a` conv of a
// block (reference model.py:246-253) reads the raw `c` output and the shortcut of the block below, forms
// y = relu(bn_c(c) + shortcut) on load (model.py:381-392), stores y for its other readers and multiplies -- no separate
// x3d_tail_fwd pass, one read of y less per block.  Resident-panel kernel, or the weights-stationary one where the layer has a
// stationary shape (stages 4 / 5); a translation unit of its own.
#include "pw_gemm_wst.h"

// called by x3d_pw_fwd (pw_fwd.hip) once the arguments are validated and `a` is filled
int pw_fwd_tail(PwGemmArgs& a, int dtype, int vec, int ovec, hipStream_t st) {
  // stages 4 / 5 (round 4): the weights-stationary kernel carries the fold too, so these layers keep their kernel
  const int shp = pw_wst_shape(a, vec, ovec);
  if (pw_wst_shape_has_tail(shp)) return dtype == X3D_F16 ? pw_wst_launch_tail<f16>(a, shp, st) : pw_wst_launch_tail<bf16>(a, shp, st);
  return dtype == X3D_F16 ? pw_bf16_launch_vec<f16, PRO_TAIL, EPI_STATS>(a, vec, ovec, st)
                          : pw_bf16_launch_vec<bf16, PRO_TAIL, EPI_STATS>(a, vec, ovec, st);
}

// ... and its one-tensor form (PRO_AFFST): v = act(s * x + t) built on load and stored -- the stem's BatchNorm + ReLU
int pw_fwd_affst(PwGemmArgs& a, int dtype, int vec, int ovec, hipStream_t st) {
  return dtype == X3D_F16 ? pw_bf16_launch_vec<f16, PRO_AFFST, EPI_STATS>(a, vec, ovec, st)
                          : pw_bf16_launch_vec<bf16, PRO_AFFST, EPI_STATS>(a, vec, ovec, st);
}
