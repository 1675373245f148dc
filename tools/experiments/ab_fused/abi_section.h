/* The C-ABI section this experiment had in include/x3d_hip.h up to X3D_ABI_VERSION 127 (removed in 128). */
/* ---- the `a` conv without its output tensor (reference model.py:305-309: a -> bn_a -> relu -> b) ------------------------
 * a_raw = W_a x is 2.25x the block input and, stored, is written once and read three times; here it is recomputed where it is
 * consumed (x3d-tf_amd/csrc/ab_fused.hip).  16-bit storage only.
 *
 * x3d_pw_gram: one pass over the conv input x [N][Cin][P]: gram [(Cin + 1)][Cin] (fp64, zero it first) += [x ; 1] x^T -- the
 *   Gram matrix and, in its last row, sum_p x; accumulated in several copies (workgroups spread their final atomics), the
 *   consumer adds them up.  Prologue (raw != NULL): x is BUILT on load, stored to `x` and then multiplied:
 *   x = relu(s1 * raw + t1 + (s2 * add + t2 | add | 0)) -- the folded residual tail of the block below (raw = its c_raw,
 *   add = its shortcut) or the stem's BatchNorm + ReLU (add == NULL).
 * x3d_bn_finalize_gram: BatchNorm (training) coefficients of a = W_a x from those sums: mean = w . sx / M,
 *   E[a^2] = w^T XX w / M with w rounded to the storage type as the matrix cores multiply it; outputs and moving-statistics
 *   update as x3d_bn_finalize. */
typedef struct {
  const void* x;               /* [N][Cin][T][H][W]: read, or (raw != NULL) written */
  const void* raw;             /* NULL | [N][Cin][P] */
  const float* raw_scale_shift;/* [Cin][2] */
  const void* add;             /* NULL | [N][Cin][P] */
  const float* add_scale_shift;/* NULL (identity) | [Cin][2] */
  double* gram;                /* x3d_pw_gram_replicas() copies of [(Cin + 1)][Cin], back to back: x3d_pw_gram_elems(Cin) doubles, += */
  int N, Cin, T, H, W, dtype;
} x3d_pw_gram_args;
long long x3d_pw_gram_elems(int Cin);
int x3d_pw_gram_replicas(void);
int x3d_pw_gram_supported(const x3d_pw_gram_args* a);
int x3d_pw_gram(const x3d_pw_gram_args* a, void* stream);
int x3d_bn_finalize_gram(const double* gram, const float* w /* [C][Cin] fp32 */, double count, const float* gamma, const float* beta,
                         float* moving_mean, float* moving_var, float eps, float momentum, int update_moving,
                         float* scale_shift, float* mean_invstd, int C, int Cin, int dtype, void* stream);

/* x3d_ab_fwd: y = depthwise3x3x3(relu(s_a * (W_a x) + t_a)), stride (1, s, s), TF-SAME padding of the ACTIVATION -- the
 * fused a -> bn_a -> relu -> b forward; y raw (pre-BN_b) as x3d_dw3d_fwd stores it, with the same statistics / SE-pool
 * epilogue (stats: the replicated layout of x3d_stats_replicas).  x3d_ab_fwd_supported(): Cin % 8 == 0, Cin <= 64, W % 8 == 0,
 * even output widths, the tile images of a workgroup within 80 KB of LDS. */
typedef struct {
  const void* x;               /* [N][Cin][T][H][W] */
  const float* a_w;            /* [C][Cin] fp32 (rounded to the storage type for the matrix cores) */
  const float* a_scale_shift;  /* [C][2] BN_a */
  const float* b_w;            /* [C][27] */
  void* y;                     /* [N][C][T][Ho][Wo] */
  double* stats;               /* NULL | BN_b statistics accumulators */
  double* pool;                /* NULL | [N][C] SE pool sums */
  int N, Cin, C, T, H, W, stride, dtype;
} x3d_ab_fwd_args;
int x3d_ab_fwd_supported(const x3d_ab_fwd_args* a);
int x3d_ab_fwd(const x3d_ab_fwd_args* a, void* stream);

