"""The ops.py wrappers of the experiment (removed from the package with ABI 128)."""
# ---- the `a` conv without its output tensor (ab_fused.hip) -----------------------------------------
def pw_gram(x, gram=None, raw=None, raw_ss=None, add=None, add_ss=None):
    """gram [(Cin + 1), Cin] (fp64) += [x ; 1] x^T over all points of x [N, Cin, T, H, W]; with `raw` the input is built on
    load (x = relu(s1 * raw + t1 + (s2 * add + t2 | add | 0))), stored into `x` and then multiplied."""
    _chk(x, gram, raw, raw_ss, add, add_ss)
    n, cin, t, h, w = x.shape
    if gram is None:     # the library's replicated layout: x3d_pw_gram_replicas() copies, summed by the consumer
        gram = torch.zeros((int(hip.load().x3d_pw_gram_replicas()), cin + 1, cin), dtype=torch.float64, device=x.device)
    a = hip.PwGramArgs(ptr(x), ptr(raw), ptr(raw_ss), ptr(add), ptr(add_ss), ptr(gram), n, cin, t, h, w, hip.dtype_code(x.dtype))
    hip.call_struct("x3d_pw_gram", a)
    return gram


def bn_finalize_gram(gram, w, count, gamma, beta, moving_mean, moving_var, eps, momentum, update_moving, scale_shift,
                     mean_invstd, dtype):
    _chk(gram, w, gamma, beta, moving_mean, moving_var, scale_shift, mean_invstd)
    c, cin = w.shape
    hip.call("x3d_bn_finalize_gram", ptr(gram), ptr(w), float(count), ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var),
             float(eps), float(momentum), int(bool(update_moving)), ptr(scale_shift), ptr(mean_invstd), c, cin,
             hip.dtype_code(dtype))


def ab_fwd_args(x, a_w, a_ss, b_w, stride, y=None, stats=None, pool=None):
    n, cin, t, h, w = x.shape
    c = a_w.shape[0]
    return hip.AbFwdArgs(ptr(x), ptr(a_w), ptr(a_ss), ptr(b_w), ptr(y), ptr(stats), ptr(pool), n, cin, c, t, h, w, stride,
                         hip.dtype_code(x.dtype))


def ab_fwd(x, a_w, a_ss, b_w, stride, y=None, stats=None, pool=None):
    """Fused a -> bn_a -> relu -> b forward (x3d_ab_fwd): y = depthwise3x3x3(relu(s * (W_a x) + t)).  stats: [C, 2] fp64
    (summed over the library's replicas on return); pool: [N, C] fp64.  Returns None when the shape is not covered."""
    _chk(x, a_w, a_ss, b_w, y, stats, pool)
    n, cin, t, h, w = x.shape
    c = a_w.shape[0]
    ho, wo = _out_hw(h, w, stride)
    if y is None:
        y = torch.empty((n, c, t, ho, wo), dtype=x.dtype, device=x.device)
    st = _Stats(stats, c)
    a = ab_fwd_args(x, a_w, a_ss, b_w, stride, y, st.arg(), pool)
    import ctypes as C
    if not hip.load().x3d_ab_fwd_supported(C.byref(a)):
        return None
    hip.call_struct("x3d_ab_fwd", a)
    st.done()
    return y


