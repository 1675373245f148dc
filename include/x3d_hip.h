/* x3d_hip.h -- C ABI of libx3d_hip.so: the MI355X (gfx950) X3D forward/backward hot path.
 *
 * The reference (fcogidi/X3D-tf) has no native/FFI layer: its boundary is the Python class
 * model.X3D plus the tf.keras layers it calls.  Every entry point below replaces one
 * group of tf.keras ops on that path; the citation after each declaration is the reference call
 * site it stands in for.  A reference maintainer binds these with ctypes.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Every pointer is DEVICE memory unless marked host.
 *  - Activations are NCTHW, contiguous, element type `dtype` (X3D_F32, X3D_BF16 or X3D_F16).  Arithmetic,
 *    weights, per-channel coefficients and reductions are fp32 (statistics accumulate in fp64); with a 16-bit
 *    `dtype` the pointwise GEMM operands are rounded to that type for the matrix cores (fp32 accumulation).
 *  - `stream` is a hipStream_t passed as void*; all work is stream-ordered, nothing allocates,
 *    nothing synchronises; functions are re-entrant.
 *  - Return value: X3D_OK, or an error code with a message in x3d_last_error() (thread local).
 *  - Accumulating outputs (stats, sums, dw, pool) are += : the caller zeroes them (one memset of
 *    its workspace per step).
 */
#ifndef X3D_HIP_H
#define X3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X3D_OK 0
#define X3D_ERR_INVALID 1
#define X3D_ERR_LAUNCH 2

#define X3D_F32 0
#define X3D_BF16 1
#define X3D_F16 2   /* IEEE half: the reference's only reduced-precision mode (Keras mixed_float16, utils.py:176-192) */

#define X3D_ACT_NONE 0
#define X3D_ACT_RELU 1
#define X3D_ACT_SWISH 2
#define X3D_ACT_SIGMOID 3

/* epilogues of x3d_pw_dgrad */
#define X3D_EPI_STORE 0       /* dx = W^T dY */
#define X3D_EPI_ADD 1         /* dx = W^T dY + add                     (identity shortcut) */
#define X3D_EPI_ADD_STRIDED 2 /* dx = W^T dY + upsample_zero(add, 2)   (shortcut conv, stride 2) */
#define X3D_EPI_SWISH_BWD 3   /* dv = (W^T dY) * swish'(gate*bn_b(braw)); per-(n,c) sums */

/* Version of THIS header: bumped with every incompatible change of a signature or struct.  x3d_version() returns the value
 * the library was built with; a binding must refuse a library whose version differs from the header it was written against
 * (x3d_tf_amd/hip.py does): a stale libx3d_hip.so would otherwise take shifted arguments silently. */
#define X3D_ABI_VERSION 133
int x3d_version(void);
const char* x3d_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  stem spatial conv: tf.pad(0,1,1) + Conv3D(k=(1,3,3), s=(1,2,2), valid, no bias)
 *     reference model.py:161-166,178-184,203-204
 *     x [N][Cin][T][H][W] -> y [N][Cout][T][Ho][Wo], Ho=(H-1)/2+1.  w [Cout][Cin][3][3] fp32.
 *     x_layout = X3D_LAYOUT_NTHWC (ABI 127): x is the caller's channels-last clip batch [N][T][H][W][Cin] itself -- the
 *     layout of the reference's input (model.py:113) -- read in place by the matrix-core kernels (16-bit storage, Cin = 3,
 *     W % 8 == 0, 16-byte aligned: x3d_stem_s_nthwc_supported); no x3d_nthwc_to_ncthw pass and no planar copy of the batch.
 * ------------------------------------------------------------------------------------------ */
#define X3D_LAYOUT_NCTHW 0
#define X3D_LAYOUT_NTHWC 1
int x3d_stem_s_nthwc_supported(int Cin, int W, int Cout, int dtype);
int x3d_stem_s_fwd(const void* x, const float* w, void* y, int N, int Cin, int T, int H, int W,
                   int Cout, int dtype, int x_layout, void* stream);
/* dW of the same conv (the input needs no gradient).  dy = grad wrt y.  dw [Cout][Cin][3][3] += */
int x3d_stem_s_wgrad(const void* x, const void* dy, float* dw, int N, int Cin, int T, int H, int W,
                     int Cout, int dtype, int x_layout, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  stem temporal depthwise conv: tf.pad(KT/2,0,0) + Conv3D(k=(KT,1,1), groups=C, no bias)
 *     reference model.py:170-175,187-194,205-206.   x,y [N][C][T][HW]; w [C][KT] fp32.
 *     stats [C][2] += (sum, sum of squares) of y as stored.
 *     INFERENCE epilogue (out_scale_shift [C][2] != NULL; stats must be NULL): y = out_act(s*conv + t) -- the stem's
 *     BatchNorm folded from the moving statistics + ReLU (model.py:196-200,207-208), so the raw conv output is not stored.
 * ------------------------------------------------------------------------------------------ */
int x3d_dwt_fwd(const void* x, const float* w, void* y, double* stats, const float* out_scale_shift, int out_act, int N,
                int C, int T, int HW, int KT, int dtype, void* stream);
/* backward of K2 through the stem's BN+ReLU: dY = A*g + B*yraw + C (coef [C][4]), g = grad wrt relu(bn(y)) masked by
 * the sign of bn(y).
 *   relu_scale_shift == NULL : g is already masked (x3d_relu_bn_bwd_reduce wrote it);
 *   relu_scale_shift [C][2]  : g is the UNMASKED gradient and the mask [s*yraw + t > 0] is applied here, on the yraw values
 *                              the kernel loads anyway -- x3d_relu_bn_bwd_reduce then runs with g == NULL (sums only) and
 *                              the masked gradient is never written or re-read (one tensor pass less each way).
 * dx [N][C][T][HW] = conv_t^T dY ; dw [C][KT] += */
int x3d_dwt_bwd(const void* g, const void* yraw, const float* relu_scale_shift, const float* coef, const void* x,
                const float* w, void* dx, float* dw, int N, int C, int T, int HW, int KT, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K1 + K2 FUSED (ABI 131): the whole stem convolution pair, reference model.py:202-206 -- conv_s, then conv_t, with nothing
 *     in between -- as one launch each way, so neither the conv_s output nor its gradient ever exists in HBM.
 *   x3d_stem_fwd: x (channels-last clip batch [N][T][H][W][3], X3D_LAYOUT_NTHWC) -> y [N][Cout][T][Ho][Wo] = conv_t(conv_s(x)),
 *     the conv_s output rounded to the storage type between the two exactly as x3d_stem_s_fwd stores it: y is bit-identical to
 *     x3d_stem_s_fwd + x3d_dwt_fwd.  stats / out_scale_shift / out_act as in x3d_dwt_fwd.
 *   x3d_stem_bwd: g (grad wrt relu(bn(y)), masked here when relu_scale_shift != NULL), yraw (= y of the forward), coef as in
 *     x3d_dwt_bwd; dw_t [Cout][KT] += , dw_s [Cout][3][3][3] += .  The conv_s output is recomputed on the matrix cores, the
 *     conv_t input gradient stays in LDS as the operand of the conv_s weight-gradient tile.
 *   x3d_stem_fused_supported: bit 0 = x3d_stem_fwd takes the shape (16-bit storage, x_layout = X3D_LAYOUT_NTHWC, Cin = 3,
 *     W % 8 == 0, Cout <= 32, KT = 5, x and y under 2 GB; x 16-byte aligned), bit 1 = x3d_stem_bwd takes it and is the faster
 *     backward (Cout <= 24; it runs, slower than K2 + K1's backward, up to 32).  Everything else runs the separate K1 / K2 entry
 *     points above.
 * ------------------------------------------------------------------------------------------ */
int x3d_stem_fused_supported(int Cin, int Cout, int KT, int N, int T, int H, int W, int dtype, int x_layout);
int x3d_stem_fwd(const void* x, const float* w_s, const float* w_t, void* y, double* stats, const float* out_scale_shift,
                 int out_act, int N, int Cin, int T, int H, int W, int Cout, int KT, int dtype, int x_layout, void* stream);
int x3d_stem_bwd(const void* g, const void* yraw, const float* relu_scale_shift, const float* coef, const void* x,
                 const float* w_s, const float* w_t, float* dw_s, float* dw_t, int N, int Cin, int T, int H, int W, int Cout,
                 int KT, int dtype, int x_layout, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3  BatchNormalization(axis=-1, eps, momentum)   reference model.py:89-92,196-199,254-257,
 *     268-271,300-303,368-371.
 *   finalize (training): stats [C][2] (sum, sumsq over `count` elements per channel) ->
 *     scale_shift [C][2] = (gamma*invstd, beta - mean*gamma*invstd), mean_invstd [C][2];
 *     if update_moving: moving = moving*momentum + batch*(1-momentum) (variance unbiased).
 *   eval_coef (inference): same outputs from the moving statistics.
 *   bwd_finalize: sums [C][2] = (sum g, sum g*yraw) -> coef [C][4] = (A,B,C,0) such that
 *     dYraw = A*g + B*yraw + C ; dgamma [C] += , dbeta [C] += .
 * ------------------------------------------------------------------------------------------ */
int x3d_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                    float* moving_mean, float* moving_var, float eps, float momentum,
                    int update_moving, float* scale_shift, float* mean_invstd, int C, void* stream);
/* REPLICATED STATISTICS.  Every `stats` accumulator of this ABI (x3d_pw_fwd, x3d_dw3d_fwd, x3d_dwt_fwd producers;
 * x3d_bn_finalize, x3d_bn_fold consumers) is x3d_stats_replicas() copies of [C][2] doubles, x3d_stats_stride(C) doubles
 * apart: a producer workgroup adds into the copy its block index selects, the consumers sum the copies.  With ONE copy
 * every workgroup of a launch ends with fp64 atomics on the same C*2 addresses (one or two L2 channels): 10-18 us per
 * forward GEMM of X3D-M measured in isolation (216->96 @ 14x14: 68.8 -> 58.4 us; 108->48 @ 28x28: 92.8 -> 77.1 us), pointwise
 * forward 4.42 -> 4.08 ms per train step.  Buffers are zeroed by the caller: x3d_stats_replicas() * x3d_stats_stride(C)
 * doubles. */
int x3d_stats_replicas(void);
long long x3d_stats_stride(int C);

/* finalize FOLDED INTO THE CONSUMER (training): the consumer of a BatchNorm whose channel is uniform per workgroup
 * (x3d_dw3d_fwd for bn_a, x3d_tail_fwd_bn for bn_c / bn_r / the stem BN) computes scale/shift from the raw statistics
 * itself -- the same arithmetic as x3d_bn_finalize, bit for bit -- and one designated workgroup per channel writes
 * scale_shift / mean_invstd (kept for the backward pass) and updates the moving statistics.  Saves one ~6 us launch
 * between producer and consumer per layer (57 per X3D-M train step). */
typedef struct {
  const double* stats;         /* [C][2] (sum, sum of squares) of the producer's raw output */
  double count;                /* elements per channel */
  const float* gamma;
  const float* beta;
  float* moving_mean;          /* updated when update_moving */
  float* moving_var;
  float eps;
  float momentum;
  int update_moving;
  float* scale_shift;          /* out [C][2] */
  float* mean_invstd;          /* out [C][2] */
} x3d_bn_fold;
/* inference coefficients (scale, shift, mean, invstd from the moving statistics): all BatchNorm layers of a model in ONE launch (inference: 84-171 layers, else one ~5 us launch each per forward).
 * `items` is an array in DEVICE memory. */
typedef struct {
  const float* gamma; const float* beta; const float* moving_mean; const float* moving_var;
  float* scale_shift;          /* out [C][2] */
  float* mean_invstd;          /* out [C][2] */
  int C;
} x3d_bn_eval_item;
int x3d_bn_eval_coef_batched(const x3d_bn_eval_item* items, int n_items, float eps, void* stream);
int x3d_bn_bwd_finalize(const double* sums, double count, const float* mean_invstd,
                        const float* gamma, float* coef, float* dgamma, float* dbeta, int C,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * K5  pointwise Conv3D(k=1, no bias) as a GEMM over points on MFMA: bottleneck a / c, shortcut
 *     `residual` (stride (1,s,s) valid), conv5.  reference model.py:246-253,292-299,360-367,80-87
 *     Input prologue (folded producer BN / SE gate / activation), applied on load:
 *       v = x ; if in_scale_shift: v = s*v + t ; if in_gate: v *= gate[n][ci] ; v = act(v)
 *     Epilogue: y stored raw; stats [Cout][2] += (sum, sumsq) of y as stored (may be NULL).
 *     INFERENCE epilogue (out_scale_shift != NULL; stats must be NULL): the BatchNorm that follows the conv -- folded from
 *     the moving statistics, model.py:300-303,368-371 -- and the residual Add + ReLU (model.py:381-392) run on the fp32
 *     accumulators, so neither the raw conv output nor a separate tail pass touches HBM:
 *       v = s_o*acc + t_o ; if out_add: v += (out_add_scale_shift ? s_r*add + t_r : add) ; y = out_act(v)
 *     (`c` conv of a block: out_add = block input (identity) or the raw shortcut conv output with bn_r as
 *     out_add_scale_shift; out_act = ReLU).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* x;               /* [N][Cin][T][H][W] */
  const float* w;              /* [Cout][Cin] fp32 */
  void* y;                     /* [N][Cout][T][Ho][Wo], Ho = ceil(H/stride) */
  double* stats;               /* [Cout][2] or NULL */
  const float* in_scale_shift; /* [Cin][2] or NULL */
  const float* in_gate;        /* [N][Cin] or NULL */
  int in_act;
  int N, Cin, Cout, T, H, W, stride, dtype;
  const void* w_panel;         /* optional (bf16 path): forward panel from x3d_pw_pack_weights, else NULL */
  /* folded residual tail (training; NULL: off).  x is then the raw `c` output of the block BELOW and the conv input is that
   * block's output, built on load and stored for its other readers -- the work of a separate x3d_tail_fwd pass:
   *   v = relu(s*x + t + (in_add_scale_shift ? s_r*in_add + t_r : in_add)) ;  in_store = v
   * (s, t) = in_scale_shift, in_act = X3D_ACT_RELU, no in_gate; 16-bit storage, stride 1: x3d_pw_fwd_tail_supported().
   * in_store without in_add: v = relu(s*x + t) -- the stem's BatchNorm + ReLU (x = the raw conv_t output, reference
   * model.py:202-210) folded into the first block's `a` conv. */
  const void* in_add;               /* [N][Cin][P]: raw shortcut-conv output of the block below, or its input (identity); NULL: no Add */
  const float* in_add_scale_shift;  /* [Cin][2] (bn_r of the block below) or NULL */
  void* in_store;                   /* [N][Cin][P] the block's output y */
  const float* out_scale_shift;     /* [Cout][2] or NULL (training form: raw store + stats) */
  const void* out_add;              /* [N][Cout][T][Ho][Wo] or NULL */
  const float* out_add_scale_shift; /* [Cout][2] applied to out_add, or NULL */
  int out_act;                      /* X3D_ACT_NONE / X3D_ACT_RELU */
} x3d_pw_fwd_args;
int x3d_pw_fwd(const x3d_pw_fwd_args* a, void* stream);
int x3d_pw_fwd_tail_supported(const x3d_pw_fwd_args* a);   /* 1: the in_add / in_store form covers this call */

/* BatchNorm-backward finalize folded into its consumers (coef_fold of the three backward argument structs below; NULL: off).
 * The coefficient table `coef` [C][4] of dYraw = A*g + B*yraw + C is x3d_bn_bwd_finalize's output: a ~6 us launch between
 * the kernel that produced the sums and the kernel that needs the table.  With coef_fold set the consuming kernel derives the
 * coefficients it needs from the sums itself, in its coefficient-table prologue (the same arithmetic, one definition:
 * the same bits), and `coef` is not read.  dgamma / dbeta non-NULL: workgroup (0, 0, 0) of the launch also adds the channel's
 * gamma / beta gradient (+=) and writes the table to coef_out -- exactly ONE launch per BatchNorm and step must be given them
 * (a data- and a weight-gradient launch that share a BatchNorm: one of the two).  x3d_pw_coef_fold_supported(): whether the
 * kernel behind a call takes it (the persistent weights-stationary kernels and the 16-bit weight-gradient kernel; the
 * entry points refuse the field otherwise). */
typedef struct {
  const double* sums;          /* [C][2] as x3d_bn_bwd_finalize takes them */
  double count;
  const float* mean_invstd;    /* [C][2] */
  const float* gamma;          /* [C] */
  float* dgamma;               /* [C] += , or NULL */
  float* dbeta;                /* [C] += , or NULL */
  float* coef_out;             /* [C][4], or NULL (written with dgamma / dbeta) */
} x3d_bn_bwd_fold;

/* data gradient: dYraw = A*g + B*yraw + C on load (coef [Cout][4]; NULL coef: dYraw = g),
 * dx = W^T dYraw with one of the X3D_EPI_* epilogues. All tensors at the conv's OUTPUT points. */
typedef struct {
  const void* g;               /* [N][Cout][P] grad wrt the BN output of this conv */
  const void* yraw;            /* [N][Cout][P] raw conv output (NULL iff coef NULL) */
  const float* coef;           /* [Cout][4] or NULL */
  const float* w;              /* [Cout][Cin] fp32 */
  void* dx;                    /* [N][Cin][P] */
  int epi;
  const void* add;             /* EPI_ADD: [N][Cin][P];  EPI_ADD_STRIDED: [N][Cin][T][ceil(H/2)][ceil(W/2)] */
  const void* braw;            /* EPI_SWISH_BWD: [N][Cin][P] raw depthwise output */
  const float* b_scale_shift;  /* EPI_SWISH_BWD: [Cin][2] */
  const float* gate;           /* EPI_SWISH_BWD: [N][Cin] or NULL */
  double* nc_sums;             /* EPI_SWISH_BWD: [N][Cin][2] += (sum dv, sum dv*braw) */
  int N, Cin, Cout, T, H, W, dtype; /* T,H,W: extents of the P = T*H*W output points */
  const void* w_panel;         /* optional (bf16 path): dgrad panel from x3d_pw_pack_weights, else NULL */
  const x3d_bn_bwd_fold* coef_fold; /* NULL | derive `coef` from the BatchNorm-backward sums (above) */
} x3d_pw_dgrad_args;
int x3d_pw_dgrad(const x3d_pw_dgrad_args* a, void* stream);

/* fused data + weight gradient of one pointwise conv (bf16 storage, <= 128 channels on either side, stride 1):
 * ONE pass over g / yraw instead of the two that x3d_pw_dgrad + x3d_pw_wgrad make, and for the `c` conv one
 * pass over braw (swish for dw and swish' for dx from the same load).  Fields as in the two structs above:
 *   epi = X3D_EPI_ADD / X3D_EPI_ADD_STRIDED : `a` conv; x = conv input [N][Cin][P], add as in dgrad
 *   epi = X3D_EPI_SWISH_BWD                 : `c` conv; the conv input is swish(gate * (s_b*braw + t_b)); x unused
 * w_panel is the DGRAD panel of x3d_pw_pack_weights (required).  x3d_pw_bwd_supported() says whether a launch
 * is covered (shape, alignment); x3d_pw_bwd fails with X3D_ERR_INVALID otherwise -- callers then use the pair. */
typedef struct {
  const void* g;               /* [N][Cout][P] */
  const void* yraw;            /* [N][Cout][P] */
  const float* coef;           /* [Cout][4] */
  const void* w_panel;         /* dgrad panel (bf16) */
  void* dx;                    /* [N][Cin][P] */
  int epi;
  const void* add;
  const void* braw;
  const float* b_scale_shift;
  const float* gate;
  double* nc_sums;
  const void* x;               /* [N][Cin][P] conv input (ADD epilogues) */
  float* dw;                   /* [Cout][Cin] += */
  int N, Cin, Cout, T, H, W, dtype;
  /* folded residual-tail backward (ADD epilogues; tail_c == NULL: off).  The conv input x is the OUTPUT y of the previous
   * residual block, so dx is the gradient that block's Add + ReLU receives (model.py:381-392): with tail_c set the
   * epilogue applies that backward instead of a separate x3d_tail_bwd pass over dx,
   *   dx = [x > 0] * (W^T dYraw + add) ;  tail_sums_c [Cin][2] += (sum dx, sum dx*tail_c) ;
   *   tail_r != NULL (the previous block has a shortcut conv): tail_sums_r [Cin][2] += (sum dx, sum dx*tail_r)
   * tail_c / tail_r: raw c-conv / shortcut-conv outputs of the previous block, [N][Cin][P]. */
  const void* tail_c;
  const void* tail_r;
  double* tail_sums_c;
  double* tail_sums_r;
  /* RECOMPUTED conv output (ADD epilogues; rc_panel == NULL: off).  The conv is followed by a training-mode BatchNorm, so
   * dYraw = A*g + B*yraw + C with yraw = W x linear in the conv input: with rc_panel set the launch streams g and x ONLY
   * (yraw / coef / dw / w_panel are not read and may be NULL) -- the caller need not keep the conv's raw output at all:
   *   dx = [W1 | M] [g ; x] + c0 (+ add, + tail)    W1 = W^T diag(A), M = W^T diag(B) W, c0 = W^T C   (x3d_pw_bwd_rc_prepare)
   *   rc_sums [Cout + 1 + Cin][Cin] += [g ; 1 ; x] x^T  over all points (zero it before the launch); x3d_pw_bwd_rc_finish turns
   *   the sums into dw = diag(A) (g x^T) + diag(B) W (x x^T) + C (sum x)^T   (reference model.py:246-257 `a` -> `bn_a`). */
  const void* rc_panel;        /* x3d_pw_bwd_rc_panel_elems(Cout, Cin) elements of the storage type, 16-byte aligned */
  const float* rc_c0;          /* [Cin] */
  float* rc_sums;              /* x3d_pw_bwd_rc_sums_elems(Cout, Cin) floats */
  /* ... of a strided shortcut conv (reference model.py:360-367; rc_panel form only, epi = X3D_EPI_STORE: dx [N][Cin][T][H][W] is
   * the gradient at the SAMPLED pixels, the operand of the `a` backward's X3D_EPI_ADD_STRIDED): x_stride = 2, x is the block
   * input [N][Cin][T][xH][xW] with H = ceil(xH / 2), W = ceil(xW / 2).  x_stride = 0 / 1: dense (x [N][Cin][T][H][W]). */
  int x_stride, xH, xW;
  /* PARTIAL weight-gradient slabs instead of fp32 atomics (dw_slab == NULL: off, dw += by atomics).  The persistent
   * weights-stationary kernels (one workgroup per CU, each with its own [Cout][Cin] partial sum) end in a flush of
   * workgroups x Cout x Cin floats; as device-scope atomics that runs at ~1.3 TB/s whatever the schedule (measured: 13-24 us of
   * a 85-105 us launch), as plain stores into a slab per workgroup at the store rate.  With dw_slab the launch writes
   * x3d_pw_bwd_dw_parts(a) slabs of Cout * Cin floats (dw is not touched) and a LATER launch adds them up in a fixed order:
   * x3d_dw_slab_reduce, or the two reduce slots of x3d_se_bnb_bwd (a small launch that is on the critical path anyway).
   * x3d_pw_bwd_dw_parts() == 0: the kernel behind this call has no slab form (leave dw_slab NULL). */
  float* dw_slab;              /* x3d_pw_bwd_dw_parts(a) * Cout * Cin floats, 16-byte aligned */
  int dw_slab_parts;           /* (ABI 132) the number of slabs dw_slab holds = what x3d_pw_bwd_dw_parts(a) returned when the buffer was
                                * sized: a launch whose grid differs (another device, another build switch between recording and
                                * replay) is refused instead of writing past the buffer or leaving slabs unwritten */
  const x3d_bn_bwd_fold* coef_fold; /* NULL | derive `coef` from the BatchNorm-backward sums (x3d_bn_bwd_fold) */
} x3d_pw_bwd_args;
int x3d_pw_bwd_supported(const x3d_pw_bwd_args* a);
int x3d_pw_bwd(const x3d_pw_bwd_args* a, void* stream);
int x3d_pw_bwd_dw_parts(const x3d_pw_bwd_args* a);
/* dw [elems] += sum over `parts` slabs of [elems] floats, parts in ascending order per element (deterministic) */
typedef struct {
  const float* slab;           /* NULL: no job */
  float* dw;
  int parts, elems;
} x3d_dw_reduce_job;
int x3d_dw_slab_reduce(const x3d_dw_reduce_job* jobs, int n_jobs, void* stream);
/* the per-step operands of the recomputed-output form.  panel_elems == 0: the layer shape is not covered (covered: Cin <= 32
 * with Cout <= 127; Cin 33..48 with Cout 65..127 or 193..223 -- the X3D stage-3 `a` convs and the first one of stage 4).  prepare: after x3d_bn_bwd_finalize produced `coef` [Cout][4]; finish: after x3d_pw_bwd, dw [Cout][Cin] +=. */
long long x3d_pw_bwd_rc_panel_elems(int Cout, int Cin);
long long x3d_pw_bwd_rc_sums_elems(int Cout, int Cin);
int x3d_pw_bwd_rc_prepare(const float* w /* [Cout][Cin] fp32 */, const float* coef, void* rc_panel, float* rc_c0, int Cout, int Cin,
                          int dtype, void* stream);
int x3d_pw_bwd_rc_finish(const float* rc_sums, const float* w, const float* coef, float* dw, int Cout, int Cin, int dtype, void* stream);
/* x3d_bn_bwd_finalize (arguments up to C: the same arithmetic, the same outputs) + x3d_pw_bwd_rc_prepare for the conv this
 * BatchNorm follows (w != NULL: its [C][Cin] weights; C <= 223) + x3d_pw_bwd_rc_finish of an EARLIER recomputed-output launch
 * (fin_sums != NULL), in ONE launch: three ~5 us launches of the backward pass's critical path become one. */
int x3d_bn_bwd_finalize_rc(const double* sums, double count, const float* mean_invstd, const float* gamma, float* coef,
                           float* dgamma, float* dbeta, int C, const float* w, void* rc_panel, float* rc_c0, int Cin,
                           const float* fin_sums, const float* fin_w, const float* fin_coef, float* fin_dw, int fin_Cout,
                           int fin_Cin, int dtype, void* stream);

/* bf16 weight panels.  The bf16 GEMMs keep their A operand (the weights) resident in LDS as bf16 rows of
 * pitch roundup(K,16)+8; without a panel every workgroup converts its rows from the fp32 master weights
 * (a latency-serialised gather that dominates small-P layers).  x3d_pw_pack_weights converts every
 * pointwise weight of the model once per step, in ONE launch, into the exact LDS image:
 *   fwd_panel   bf16 [roundup(Cout,32)][roundup(Cin,16)+8]   row co, column ci   (zero padded)
 *   dgrad_panel bf16 [roundup(Cin,32)][roundup(Cout,16)+8]   row ci, column co   (may be NULL)
 * each followed by a second image of the same matrix tiled as the 32x32x16 MFMA A operand,
 *   bf16 [rows/32][roundup(cols,16)/16][64 lanes][8]: lane (r = lane%32, half = lane/32) holds W[32*mi + r][16*ks + 8*half ..+7]
 * (one contiguous 1 KB wave load per k-step: the weights-stationary stage-5 kernel keeps it in registers).
 * x3d_pw_panel_elems(rows, cols) is the element count of BOTH images; always size panels with it.
 * `items` is an array in DEVICE memory; x3d_pw_panel_elems gives the element count of a panel. */
typedef struct {
  const float* w;              /* [Cout][Cin] fp32 */
  void* fwd_panel;
  void* dgrad_panel;
  int Cout, Cin;
} x3d_pw_pack_item;
long long x3d_pw_panel_elems(int rows, int cols);
int x3d_pw_pack_weights(const x3d_pw_pack_item* items, int n_items, int dtype /* X3D_BF16 or X3D_F16: element type of the panels */, void* stream);

/* weight gradient: dw [Cout][Cin] += sum_{n,p} dYraw[n][co][p] * act_in(x)[n][ci][q(p)] */
typedef struct {
  const void* g;
  const void* yraw;
  const float* coef;           /* as in dgrad */
  const void* x;               /* conv input [N][Cin][T][H][W] */
  const float* in_scale_shift; /* prologue of the forward pass, replayed */
  const float* in_gate;
  int in_act;
  float* dw;                   /* [Cout][Cin] += */
  int N, Cin, Cout, T, H, W, stride, dtype; /* T,H,W: INPUT extents (as in fwd) */
  float* dw_slab;              /* NULL | partial slabs instead of atomics, as x3d_pw_bwd_args.dw_slab: x3d_pw_wgrad_dw_parts(a) * Cout * Cin
                                * floats, every one of them written (dw untouched); added up by x3d_dw_slab_reduce / x3d_se_bnb_bwd */
  int dw_slab_parts;           /* the number of slabs dw_slab holds (x3d_pw_wgrad_dw_parts(a) when it was sized): checked against the grid */
  const x3d_bn_bwd_fold* coef_fold; /* NULL | derive `coef` from the BatchNorm-backward sums (x3d_bn_bwd_fold) */
} x3d_pw_wgrad_args;
int x3d_pw_wgrad(const x3d_pw_wgrad_args* a, void* stream);
int x3d_pw_wgrad_dw_parts(const x3d_pw_wgrad_args* a);   /* 0: the kernel behind this call has no slab form */

/* Which kernel instantiation a pointwise launch with these arguments runs ("pw_gemm_wst_kernel<1, 100, 6, 1, 27, 1>", ...),
 * without launching anything: the dispatch of x3d_pw_fwd / x3d_pw_dgrad / x3d_pw_wgrad / x3d_pw_bwd in dry-run mode (works
 * without a GPU; pointers are only inspected for alignment).  Exactly one struct is non-NULL.  Used by the dispatch-coverage
 * test (every instantiation the BASELINE configurations launch has an oracle-parity case) and by the profiling tools. */
int x3d_pw_kernel_name(const x3d_pw_fwd_args* fwd, const x3d_pw_dgrad_args* dgrad, const x3d_pw_wgrad_args* wgrad,
                       const x3d_pw_bwd_args* bwd, char* out, int cap);
/* 1: the kernel this call dispatches to takes coef_fold (x3d_bn_bwd_fold).  Exactly one struct is non-NULL. */
int x3d_pw_coef_fold_supported(const x3d_pw_dgrad_args* dgrad, const x3d_pw_wgrad_args* wgrad, const x3d_pw_bwd_args* bwd);

/* ------------------------------------------------------------------------------------------
 * K6  channelwise Conv3D(k=(3,3,3), s=(1,s,s), padding='same', groups=C, no bias)
 *     reference model.py:259-267.  TF-SAME (asymmetric) padding.  The HBM-bound headline kernel.
 *     Prologue on load: v = act(s*x + t) inside the image, 0 in the padding.
 *     Epilogue: stats [C][2] += ; pool [N][C] += sum over T,Ho,Wo of y (SE squeeze, model.py:277,312)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* x;               /* [N][C][T][H][W] */
  const float* w;              /* [C][3][3][3] fp32 */
  void* y;                     /* [N][C][T][Ho][Wo] */
  const float* in_scale_shift; /* [C][2] or NULL */
  int in_act;
  double* stats;               /* [C][2] or NULL */
  double* pool;                /* [N][C] or NULL */
  int N, C, T, H, W, stride, dtype;
  const x3d_bn_fold* in_bn;    /* NULL, or: the prologue's scale/shift come from these statistics (in_scale_shift unused) */
} x3d_dw3d_fwd_args;
int x3d_dw3d_fwd(const x3d_dw3d_fwd_args* a, void* stream);

/* fused backward: dB = A*dv + B*braw + C with per-(n,c) coef_nc [N][C][4];
 *   ga = (conv^T dB) * [s_a*araw + t_a > 0]           -> [N][C][T][H][W]
 *   a_sums [C][2] += (sum ga, sum ga*araw) ; dw [C][27] += sum dB * relu(bn_a(araw))_padded */
typedef struct {
  const void* dv;              /* [N][C][T][Ho][Wo] */
  const void* braw;            /* [N][C][T][Ho][Wo] */
  const float* coef_nc;        /* [N][C][4] */
  const void* araw;            /* [N][C][T][H][W] */
  const float* a_scale_shift;  /* [C][2] */
  const float* w;              /* [C][27] */
  void* ga;                    /* [N][C][T][H][W] */
  double* a_sums;              /* [C][2] */
  float* dw;                   /* [C][27] */
  int N, C, T, H, W, stride, dtype;
} x3d_dw3d_bwd_args;
int x3d_dw3d_bwd(const x3d_dw3d_bwd_args* a, void* stream);
/* Which kernel instantiation a launch with these arguments runs ("dw3d_fwd_kernel<bf16, S, SW, NSV, CV>"),
 * without launching anything: the dispatch of x3d_dw3d_fwd / x3d_dw3d_bwd in dry-run mode.  Used by
 * bench.py so that its roofline row names the same kernel as the rocprofv3 summary.  Exactly one of
 * fwd / bwd is non-NULL.  Returns X3D_OK and a NUL-terminated string in out[0..cap). */
int x3d_dw3d_kernel_name(const x3d_dw3d_fwd_args* fwd, const x3d_dw3d_bwd_args* bwd, char* out, int cap);

/* ------------------------------------------------------------------------------------------
 * K7/K8  squeeze-excite MLP: se_pool -> se_fc1(+bias, ReLU) -> se_fc2(+bias, sigmoid)
 *     reference model.py:274-290,311-315.  pool_sums [N][C] = sum over P points of braw (from K6);
 *     pooled = s_b * pool_sums / P + t_b.  gate [N][C], hidden [N][Wd] (saved for backward).
 * ------------------------------------------------------------------------------------------ */
int x3d_se_fwd(const double* pool_sums, double P, const float* b_scale_shift, const float* w1,
               const float* b1, const float* w2, const float* b2, float* gate, float* hidden, int N,
               int C, int Wd, void* stream);
/* backward of the SE branch + BN_b, from the per-(n,c) sums of x3d_pw_dgrad(EPI_SWISH_BWD):
 *   nc_sums [N][C][2] = (S1 = sum dv, S2 = sum dv*braw)
 *   dgate = s_b*S2 + t_b*S1 -> sigmoid' -> fc2^T -> relu' -> fc1^T -> dpool [N][C]
 *   dw1,db1,dw2,db2 += ; then BN_b backward over du = dv*gate + dpool/P:
 *   coef_nc [N][C][4] = (A*gate, B, C + A*dpool/P, 0) ; dgamma_b, dbeta_b += .
 * Without SE (w1 == NULL): gate = 1, dpool = 0. */
typedef struct {
  const double* nc_sums;       /* [N][C][2] */
  const double* pool_sums;     /* [N][C] (forward) or NULL without SE */
  double P;
  const float* b_scale_shift;  /* [C][2] */
  const float* b_mean_invstd;  /* [C][2] */
  const float* gamma_b;        /* [C] */
  const float* w1; const float* b1; const float* w2; const float* b2; /* SE params or NULL */
  const float* gate;           /* [N][C] or NULL */
  const float* hidden;         /* [N][Wd] or NULL */
  float* dw1; float* db1; float* dw2; float* db2;
  float* dgamma_b; float* dbeta_b;
  float* coef_nc;              /* out [N][C][4] */
  float* scratch;              /* N*(2*C + Wd) floats: dpool [N][C] | dz2 [N][C] | dz1 [N][Wd] */
  int N, C, Wd;
  x3d_dw_reduce_job reduce[2]; /* weight-gradient slabs of earlier x3d_pw_bwd launches, added up by extra workgroups of this launch */
} x3d_se_bnb_bwd_args;
int x3d_se_bnb_bwd(const x3d_se_bnb_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9  residual tail: Add + ReLU  (reference model.py:381-392)
 *     y = relu(s_c*c + t_c + shortcut), shortcut = s_r*r + t_r (conv shortcut) or x (identity);
 *     shortcut == NULL: y = relu(s_c*c + t_c)  (BN + ReLU after the stem, model.py:207-208)
 * ------------------------------------------------------------------------------------------ */
/* the same with the BatchNorm finalize of bn_c (and bn_r) folded in: see x3d_bn_fold */
int x3d_tail_fwd_bn(const void* c_raw, const x3d_bn_fold* c_bn, const void* shortcut, const x3d_bn_fold* r_bn,
                    void* y, int N, int C, long long P, int dtype, void* stream);
int x3d_tail_fwd(const void* c_raw, const float* c_scale_shift, const void* shortcut,
                 const float* r_scale_shift /* NULL: identity */, void* y, int N, int C, long long P,
                 int dtype, void* stream);
/* g = dy * [y > 0] written in place over dy; sums_c [C][2] += (sum g, sum g*c_raw);
 * if r_raw: sums_r [C][2] += (sum g, sum g*r_raw) */
int x3d_tail_bwd(void* dy_g, const void* y, const void* c_raw, const void* r_raw, double* sums_c,
                 double* sums_r, int N, int C, long long P, int dtype, void* stream);
/* generic ReLU+BN backward reduce for stem / conv5: z = s*yraw + t.
 *   dy != NULL : g = dy * [z > 0]                (g may alias dy)
 *   dy == NULL : g = dpool[n][c] / P * [z > 0]   (global-average-pool backward, model.py:94,118)
 * sums [C][2] += (sum g, sum g*yraw).  g == NULL: sums only (the consumer applies the mask: x3d_dwt_bwd). */
int x3d_relu_bn_bwd_reduce(const void* dy, const float* dpool, const void* yraw,
                           const float* scale_shift, void* g, double* sums, int N, int C,
                           long long P, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K5s even-pixel copy of a block input (ABI 133): dst [planes][ceil(H/2)][ceil(W/2)] = src [planes][H][W] at the even rows and
 *     columns, planes = N*C*T -- the pixels the stride-(1,2,2) 'valid' shortcut conv samples (reference model.py:360-367).
 *     With it the shortcut conv's forward, data gradient and weight gradient run as dense launches of x3d_pw_fwd /
 *     x3d_pw_bwd / x3d_pw_wgrad on the compact tensor (stride = 1, x_stride = 0) instead of the strided gathers.
 * ------------------------------------------------------------------------------------------ */
int x3d_subsample2(const void* src, void* dst, long long planes, int H, int W, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K7/K10  head: pool5 (GlobalAveragePooling3D of relu(bn(conv5))), fc1 (1x1x1 conv on the pooled
 *     vector, no bias, ReLU), Dropout, fc2 Dense(+bias), Softmax(fp32), view mean.
 *     reference model.py:94-111,117-127
 * ------------------------------------------------------------------------------------------ */
int x3d_pool_fwd(const void* x_raw, const float* scale_shift, float* pooled /* [N][C] */, int N, int C,
                 long long P, int dtype, void* stream);
/* y[n][m] = act( sum_k (x[n][k] * (mask ? mask[n][k]*mask_scale : 1)) * w[m][k] + b[m] ), all fp32 */
int x3d_dense_fwd(const float* x, const float* mask, float mask_scale, const float* w, const float* b,
                  float* y, int act, int N, int K, int M, void* stream);
/* dy is grad wrt the activated output y (ReLU only).  dx [N][K] (may be NULL), dw [M][K] +=, db [M] += */
int x3d_dense_bwd(const float* dy, const float* y, int act, const float* x, const float* mask,
                  float mask_scale, const float* w, float* dx, float* dw, float* db, int N, int K,
                  int M, void* stream);
/* probs = softmax(logits); loss_rows[n] = -log q_y + log sum_j q_j with q = clip(p,1e-7,1-1e-7)
 * (tf.keras SparseCategoricalCrossentropy on probabilities, train.py:104); dlogits = d(mean loss)/dlogits
 * scaled by grad_scale (= 1/global_batch).  labels int32; loss_rows / dlogits may be NULL.  A label outside
 * [0, M) is never used as an index: its loss row is NaN and its dlogits row zero. */
int x3d_softmax_xent(const float* logits, const int* labels, float* probs, float* loss_rows,
                     float* dlogits, float grad_scale, int N, int M, void* stream);
/* out[v][m] = mean over `views` consecutive rows (model.py:123-126) */
int x3d_view_mean(const float* probs, float* out, int videos, int views, int M, void* stream);

/* ------------------------------------------------------------------------------------------
 * K12  SGD with Nesterov momentum + L2 (train.py:89-92, model.py:47):
 *     g' = g + 2*wd*w (where l2_mask) ; v = m*v - lr*g' ; w = w + m*v - lr*g'
 *     flat fp32 arrays of n elements; wd_mask [n] in {0,1} as uint8 (NULL: no decay)
 * ------------------------------------------------------------------------------------------ */
int x3d_sgd_nesterov(float* w, float* v, const float* g, const unsigned char* l2_mask, float lr,
                     float momentum, float weight_decay, float grad_scale, long long n, void* stream);
/* Adam, the reference's other optimizer branch (tf.optimizers.Adam(learning_rate), train.py:93-95; Keras defaults
 * beta1 0.9, beta2 0.999, eps 1e-7):  g' as above; m = b1*m + (1-b1)*g'; v = b2*v + (1-b2)*g'^2;
 * w -= lr * sqrt(1 - b2^step) / (1 - b1^step) * m / (sqrt(v) + eps).  `step` counts from 1. */
int x3d_adam(float* w, float* m, float* v, const float* g, const unsigned char* l2_mask, float lr, float beta1,
             float beta2, float eps, float weight_decay, float grad_scale, long long step, long long n, void* stream);
/* LossScaleOptimizer support (Keras mixed_float16, train.py:99-100): *flag (device int the caller set to 1) is cleared
 * when any of the n values is inf / nan -- the step is then skipped and the loss scale halved. */
int x3d_all_finite(const float* g, long long n, int* flag, void* stream);
/* sum of squares of the masked entries (L2 regularisation loss term), out [1] double += */
int x3d_l2_sumsq(const float* w, const unsigned char* l2_mask, double* out, long long n, void* stream);

/* layout helpers at the module boundary: NTHWC (reference, model.py:113) <-> NCTHW (internal) */
int x3d_nthwc_to_ncthw(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C,
                       long long P, void* stream);

/* ------------------------------------------------------------------------------------------
 * eval-side view construction: decoded video -> the views x crops clips eval.py feeds the model
 *     reference transforms.py:48-65 (temporal looping sampler), :112-147 (short side -> `size`, bilinear, cast
 *     back to uint8), :149-190 (uniform crop, ceil offsets), utils.py:42-72 (x/255 - mean, / std),
 *     dataloader.py:107-116 (clip order: crops major, then views).
 *     video [F][H][W][3] uint8 (device) -> out [crops*views][T][size][size][3] (X3D_F32 / X3D_BF16), channels-last
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const unsigned char* video;
  void* out;
  int F, H, W;
  int T, views, crops, size;
  float mean[3];
  float std[3];
  int dtype;
} x3d_eval_views_args;
int x3d_eval_views(const x3d_eval_views_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * train-side clip construction (the device half of the training input pipeline; decoding stays on
 * the host): decoded video -> one augmented training clip.
 *     reference transforms.py:31-47 (random start, every `rate`-th frame, the video looped), :112-147
 *     (random_short_side_resize: short side -> int(jitter), long side floor((long/short) * jitter) in float32,
 *     bilinear, cast back to uint8), :199-203 (tf.image.random_crop: one (y0, x0) for every frame of the clip),
 *     :205-206 (flip_left_right on EVERY training clip: `random_hflip` is just `is_training`, dataloader.py:136),
 *     utils.py:42-72 (x/255 - mean, / std).
 *     The random draws (start, jitter, y0, x0) are the caller's: TF's generator streams are not reproduced [TF-3p].
 *     video [F][H][W][3] uint8 (device) -> out [T][size][size][3] (X3D_F32 / X3D_BF16), channels-last
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const unsigned char* video;
  void* out;
  int F, H, W;
  int T, rate, start;          /* frame j of the clip = (start + j * rate) mod F */
  float jitter;                /* short-side target, uniform in [TRAIN_JITTER_SCALES) */
  int size, y0, x0;            /* crop size and offsets inside the resized frame */
  int flip;                    /* 1: mirror left-right (the reference always does in training) */
  float mean[3];
  float std[3];
  int dtype;
} x3d_train_clip_args;
int x3d_train_clip(const x3d_train_clip_args* a, void* stream);
/* extents of a H x W frame after random_short_side_resize with target `jitter` (transforms.py:126-141) */
int x3d_train_resized_hw(int H, int W, float jitter, int* new_h, int* new_w);

/* ------------------------------------------------------------------------------------------
 * host helper: CRC32C (Castagnoli), the checksum of TF tensor-bundle checkpoints
 *     (reference train.py:151-158 / utils.py restore path reads such files through tf.train.Checkpoint)
 * ------------------------------------------------------------------------------------------ */
uint32_t x3d_crc32c(const void* data, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* X3D_HIP_H */
