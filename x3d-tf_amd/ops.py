"""Tensor-level wrappers over the C ABI: shape inference + pointer plumbing, nothing else.

Every function launches hand-written HIP kernels from libx3d_hip.so on the current stream.  Tensors
are NCTHW activations (fp32 or bf16) and fp32 parameters / coefficient vectors, all on the GPU.
"""
import ctypes

import torch

from . import hip
from .hip import (ACT_NONE, ACT_RELU, ACT_SWISH, EPI_ADD, EPI_ADD_STRIDED, EPI_STORE,  # noqa: F401
                  EPI_SWISH_BWD, ptr)


def _chk(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise hip.X3DHipError("x3d ops need GPU tensors (no CPU fallback)")
            if not t.is_contiguous():
                raise hip.X3DHipError("x3d ops need contiguous tensors")


def _out_hw(h, w, stride):
    return -(-h // stride), -(-w // stride)


# ---- stem ---------------------------------------------------------------------------------------
def stem_s_fwd(x, w, y=None, channels_last=False):
    """channels_last: x is the clip batch [N, T, H, W, Cin] as the reference's model takes it (X3D_LAYOUT_NTHWC)."""
    _chk(x, w, y)
    if channels_last:
        n, t, h, ww, cin = x.shape
    else:
        n, cin, t, h, ww = x.shape
    cout = w.shape[0]
    ho, wo = (h - 1) // 2 + 1, (ww - 1) // 2 + 1
    if y is None:
        y = torch.empty((n, cout, t, ho, wo), dtype=x.dtype, device=x.device)
    hip.call("x3d_stem_s_fwd", ptr(x), ptr(w), ptr(y), n, cin, t, h, ww, cout, hip.dtype_code(x.dtype), int(channels_last))
    return y


def stem_s_wgrad(x, dy, dw, channels_last=False):
    _chk(x, dy, dw)
    if channels_last:
        n, t, h, ww, cin = x.shape
    else:
        n, cin, t, h, ww = x.shape
    hip.call("x3d_stem_s_wgrad", ptr(x), ptr(dy), ptr(dw), n, cin, t, h, ww, dy.shape[1],
             hip.dtype_code(x.dtype), int(channels_last))


# ---- replicated statistics accumulators (include/x3d_hip.h) ----------------------------------------
def stats_buffer(c, device):
    """Zeroed accumulator in the library's replicated layout for c channels."""
    r, stride = hip.stats_layout(c)
    return torch.zeros(r * stride, dtype=torch.float64, device=device)


def stats_sum(buf, c):
    """[c, 2] totals of a replicated accumulator."""
    r, stride = hip.stats_layout(c)
    return buf.view(r, stride)[:, :2 * c].sum(0).view(c, 2)


_KEEP = []


class _Stats:
    """Producers called with a plain [C, 2] tensor (tests, tools) run on a temporary replicated accumulator whose
    totals are added to the tensor afterwards; a tensor already in the replicated layout is passed through."""

    def __init__(self, user, c):
        self.user, self.c, self.tmp = user, c, None
        if user is not None and user.numel() == 2 * c:
            self.tmp = stats_buffer(c, user.device)

    def arg(self):
        return self.user if self.tmp is None else self.tmp

    def done(self):
        if self.tmp is not None:
            self.user += stats_sum(self.tmp, self.c).view_as(self.user)


def _expand_stats(stats, c):
    """Consumers called with plain [C, 2] totals: replica 0 holds them, the others are zero."""
    if stats.numel() != 2 * c:
        return stats
    buf = stats_buffer(c, stats.device)
    buf[:2 * c] = stats.reshape(-1)
    return buf


def dwt_fwd(x, w, y=None, stats=None, out_ss=None, out_act=ACT_NONE):
    """out_ss [C][2]: the inference epilogue y = out_act(s*conv + t) (no statistics)."""
    _chk(x, w, y, stats, out_ss)
    n, c, t, h, ww = x.shape
    if y is None:
        y = torch.empty_like(x)
    st = _Stats(stats, c)
    hip.call("x3d_dwt_fwd", ptr(x), ptr(w), ptr(y), ptr(st.arg()), ptr(out_ss), out_act, n, c, t, h * ww, w.shape[1],
             hip.dtype_code(x.dtype))
    st.done()
    return y


def dwt_bwd(g, yraw, coef, x, w, dx, dw, relu_ss=None):
    """relu_ss [C][2]: g is the unmasked gradient, the ReLU mask of bn(yraw) is applied inside the kernel."""
    _chk(g, yraw, coef, x, w, dx, dw, relu_ss)
    n, c, t, h, ww = x.shape
    hip.call("x3d_dwt_bwd", ptr(g), ptr(yraw), ptr(relu_ss), ptr(coef), ptr(x), ptr(w), ptr(dx), ptr(dw), n, c, t,
             h * ww, w.shape[1], hip.dtype_code(x.dtype))


def stem_fused_supported(x, cout, kt=5):
    """x: the channels-last clip batch [N, T, H, W, Cin].  x3d_stem_fused_supported: bit 0 = stem_fwd takes the shape, bit 1 =
    stem_bwd takes it and is the faster backward."""
    n, t, h, ww, cin = x.shape
    code = hip.dtype_code(x.dtype)
    return int(hip.load().x3d_stem_fused_supported(cin, cout, kt, n, t, h, ww, code, 1))


def stem_fwd(x, w_s, w_t, y=None, stats=None, out_ss=None, out_act=ACT_NONE):
    """conv_s -> conv_t in one launch (x3d_stem_fwd): x is the channels-last clip batch [N, T, H, W, 3]; the conv_s output
    never reaches HBM.  stats / out_ss / out_act as in dwt_fwd."""
    _chk(x, w_s, w_t, y, stats, out_ss)
    n, t, h, ww, cin = x.shape
    cout = w_s.shape[0]
    ho, wo = (h - 1) // 2 + 1, (ww - 1) // 2 + 1
    if y is None:
        y = torch.empty((n, cout, t, ho, wo), dtype=x.dtype, device=x.device)
    st = _Stats(stats, cout)
    hip.call("x3d_stem_fwd", ptr(x), ptr(w_s), ptr(w_t), ptr(y), ptr(st.arg()), ptr(out_ss), out_act, n, cin, t, h, ww, cout,
             w_t.shape[1], hip.dtype_code(x.dtype), 1)
    st.done()
    return y


def stem_bwd(g, yraw, coef, x, w_s, w_t, dw_s, dw_t, relu_ss=None):
    """Backward of the fused stem (x3d_stem_bwd): dw_s, dw_t += ; the conv_s output is recomputed, its gradient stays on chip."""
    _chk(g, yraw, coef, x, w_s, w_t, dw_s, dw_t, relu_ss)
    n, t, h, ww, cin = x.shape
    hip.call("x3d_stem_bwd", ptr(g), ptr(yraw), ptr(relu_ss), ptr(coef), ptr(x), ptr(w_s), ptr(w_t), ptr(dw_s), ptr(dw_t),
             n, cin, t, h, ww, w_s.shape[0], w_t.shape[1], hip.dtype_code(x.dtype), 1)


# ---- batch norm ---------------------------------------------------------------------------------
def bn_finalize(stats, count, gamma, beta, mmean, mvar, eps, momentum, update, ss, mi):
    _chk(stats, gamma, beta, mmean, mvar, ss, mi)
    stats = _expand_stats(stats, gamma.numel())
    hip.call("x3d_bn_finalize", ptr(stats), float(count), ptr(gamma), ptr(beta), ptr(mmean), ptr(mvar),
             float(eps), float(momentum), int(update), ptr(ss), ptr(mi), gamma.numel())


def bn_fold(stats, count, gamma, beta, mmean, mvar, eps, momentum, update, ss, mi):
    """x3d_bn_fold for a consumer that runs the finalize itself (x3d_dw3d_fwd in_bn / x3d_tail_fwd_bn)."""
    _chk(stats, gamma, beta, mmean, mvar, ss, mi)
    stats = _expand_stats(stats, gamma.numel())
    _KEEP.append(stats)   # the struct holds a raw pointer
    del _KEEP[:-64]
    return hip.BnFold(ptr(stats), float(count), ptr(gamma), ptr(beta), ptr(mmean), ptr(mvar), float(eps), float(momentum),
                      int(update), ptr(ss), ptr(mi))


def bn_eval_coef(gamma, beta, mmean, mvar, eps, ss, mi):
    _chk(gamma, beta, mmean, mvar, ss, mi)
    item = hip.BnEvalItem(ptr(gamma), ptr(beta), ptr(mmean), ptr(mvar), ptr(ss), ptr(mi), gamma.numel())
    table = torch.frombuffer(bytearray(bytes(item)), dtype=torch.uint8).to(gamma.device)   # (the item table lives in device memory)
    hip.call("x3d_bn_eval_coef_batched", table.data_ptr(), 1, float(eps))
    torch.cuda.current_stream().synchronize()     # `table` is freed on return


def bn_bwd_finalize(sums, count, mi, gamma, coef, dgamma, dbeta):
    _chk(sums, mi, gamma, coef, dgamma, dbeta)
    hip.call("x3d_bn_bwd_finalize", ptr(sums), float(count), ptr(mi), ptr(gamma), ptr(coef),
             ptr(dgamma), ptr(dbeta), gamma.numel())


# ---- pointwise ----------------------------------------------------------------------------------
def pw_pack_weights(weights, dgrad=True, dtype=torch.bfloat16):
    """fp32 [Cout, Cin] weights -> list of (fwd_panel, dgrad_panel) LDS-image panels of the 16-bit storage type `dtype`,
    one launch (x3d_pw_pack_weights)."""
    import ctypes as C
    lib = hip.load()
    items = (hip.PwPackItem * len(weights))()
    out = []
    for i, w in enumerate(weights):
        _chk(w)
        cout, cin = w.shape
        fp = torch.empty(lib.x3d_pw_panel_elems(cout, cin), dtype=dtype, device=w.device)
        dp = torch.empty(lib.x3d_pw_panel_elems(cin, cout), dtype=dtype, device=w.device) if dgrad else None
        items[i] = hip.PwPackItem(ptr(w), ptr(fp), ptr(dp), cout, cin)
        out.append((fp, dp))
    table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(weights[0].device)
    hip.call("x3d_pw_pack_weights", table.data_ptr(), len(weights), hip.dtype_code(dtype))
    torch.cuda.current_stream().synchronize()   # `table` must outlive the launch
    return out


def pw_fwd(x, w, y=None, stats=None, in_ss=None, in_gate=None, in_act=ACT_NONE, stride=1, w_panel=None,
           out_ss=None, out_add=None, out_add_ss=None, out_act=ACT_NONE, in_add=None, in_add_ss=None, in_store=None):
    """out_ss [Cout][2]: the inference epilogue y = out_act(s_o*acc + t_o [+ s_r*out_add + t_r]) (no statistics).
    in_add / in_add_ss / in_store: the residual tail of the block below folded into the prologue (x = its raw c output)."""
    _chk(x, w, y, stats, in_ss, in_gate, w_panel, out_ss, out_add, out_add_ss, in_add, in_add_ss, in_store)
    n, cin, t, h, ww = x.shape
    cout = w.shape[0]
    ho, wo = _out_hw(h, ww, stride)
    if y is None:
        y = torch.empty((n, cout, t, ho, wo), dtype=x.dtype, device=x.device)
    st = _Stats(stats, cout)
    a = hip.PwFwdArgs(ptr(x), ptr(w), ptr(y), ptr(st.arg()), ptr(in_ss), ptr(in_gate), in_act, n, cin,
                      cout, t, h, ww, stride, hip.dtype_code(x.dtype), ptr(w_panel), in_add=ptr(in_add),
                      in_add_scale_shift=ptr(in_add_ss), in_store=ptr(in_store), out_scale_shift=ptr(out_ss),
                      out_add=ptr(out_add), out_add_scale_shift=ptr(out_add_ss), out_act=out_act)
    hip.call_struct("x3d_pw_fwd", a)
    st.done()
    return y


def bn_bwd_fold(sums, count, mi, gamma, dgamma=None, dbeta=None, coef_out=None):
    """x3d_bn_bwd_fold for the `coef_fold` argument of pw_dgrad / pw_wgrad / pw_bwd: the consumer derives its coefficient table
    from the BatchNorm-backward sums; dgamma / dbeta / coef_out: this launch also publishes them (one launch per BatchNorm)."""
    _chk(sums, mi, gamma, dgamma, dbeta, coef_out)
    f = hip.BnBwdFold(ptr(sums), float(count), ptr(mi), ptr(gamma), ptr(dgamma), ptr(dbeta), ptr(coef_out))
    _KEEP.append((f, sums, mi, gamma, dgamma, dbeta, coef_out))
    del _KEEP[:-64]
    return f


def pw_dgrad(g, yraw, coef, w, dx, epi=EPI_STORE, add=None, braw=None, b_ss=None, gate=None,
             nc_sums=None, w_panel=None, coef_fold=None):
    """g/yraw: [N,Cout,T,H,W]; dx: [N,Cin,T,H,W]."""
    _chk(g, yraw, coef, w, dx, add, braw, b_ss, gate, nc_sums)
    n, cout, t, h, ww = g.shape
    cin = w.shape[1]
    a = hip.PwDgradArgs(ptr(g), ptr(yraw), ptr(coef), ptr(w), ptr(dx), epi, ptr(add), ptr(braw),
                        ptr(b_ss), ptr(gate), ptr(nc_sums), n, cin, cout, t, h, ww,
                        hip.dtype_code(g.dtype), ptr(w_panel))
    if coef_fold is not None:
        a.coef_fold = hip.fold_address(coef_fold)
    hip.call_struct("x3d_pw_dgrad", a)
    return dx


def pw_bwd(g, yraw, coef, w_panel, dx, dw, epi, x=None, add=None, braw=None, b_ss=None, gate=None, nc_sums=None,
           tail_c=None, tail_r=None, tail_sums_c=None, tail_sums_r=None, slab=False, coef_fold=None):
    """Fused dgrad + wgrad (x3d_pw_bwd).  g/yraw: [N,Cout,T,H,W]; dx: [N,Cin,T,H,W]; dw: [Cout,Cin] +=.
    tail_c (/ tail_r) + their [Cin, 2] fp64 sums: the folded Add + ReLU backward of the block whose output is x.
    slab: the weight gradient through per-workgroup partial slabs + x3d_dw_slab_reduce instead of fp32 atomics (None is
    returned when the kernel behind the call has no slab form).
    Returns False (nothing launched) when the fused kernel does not cover the call."""
    _chk(g, yraw, coef, w_panel, dx, dw, x, add, braw, b_ss, gate, nc_sums, tail_c, tail_r, tail_sums_c, tail_sums_r)
    n, cout, t, h, ww = g.shape
    cin = dx.shape[1]
    a = hip.PwBwdArgs(ptr(g), ptr(yraw), ptr(coef), ptr(w_panel), ptr(dx), epi, ptr(add), ptr(braw), ptr(b_ss),
                      ptr(gate), ptr(nc_sums), ptr(x), ptr(dw), n, cin, cout, t, h, ww, hip.dtype_code(g.dtype),
                      ptr(tail_c), ptr(tail_r), ptr(tail_sums_c), ptr(tail_sums_r))
    if coef_fold is not None:
        a.coef_fold = hip.fold_address(coef_fold)
    import ctypes as C
    if not hip.load().x3d_pw_bwd_supported(C.byref(a)):
        return False
    if slab:
        parts = int(hip.load().x3d_pw_bwd_dw_parts(C.byref(a)))
        if parts <= 0:
            return None
        buf = torch.full((parts * cout * cin,), float("nan"), dtype=torch.float32, device=g.device)   # (every slab element must be written)
        a.dw_slab, a.dw_slab_parts = ptr(buf), parts
        hip.call_struct("x3d_pw_bwd", a)
        dw_slab_reduce([(buf, dw, parts)])
        return True
    hip.call_struct("x3d_pw_bwd", a)
    return True


def dw_reduce_jobs(jobs):
    """[(slab [parts * elems] fp32, dw [elems...] fp32, parts), ...] -> a ctypes array of x3d_dw_reduce_job"""
    arr = (hip.DwReduceJob * max(len(jobs), 1))()
    for i, (slab_, dw_, parts) in enumerate(jobs):
        _chk(slab_, dw_)
        assert slab_.numel() == parts * dw_.numel()
        arr[i] = hip.DwReduceJob(ptr(slab_), ptr(dw_), parts, dw_.numel())
    return arr


def dw_slab_reduce(jobs):
    """dw += sum of its partial slabs (x3d_dw_slab_reduce): one or two jobs."""
    arr = dw_reduce_jobs(jobs)
    hip.call("x3d_dw_slab_reduce", arr, len(jobs))


def pw_bwd_rc(g, x, w, coef, dx, dw, epi, add, tail_c=None, tail_r=None, tail_sums_c=None, tail_sums_r=None, x_stride=1):
    """Fused dgrad + wgrad of an `a` conv WITHOUT its raw output (x3d_pw_bwd with rc_panel: the conv output y = W x is
    folded algebraically into the BatchNorm backward dY = A g + B y + C).  Three launches: x3d_pw_bwd_rc_prepare (per-step panel
    [W^T diag(A) | W^T diag(B) W] and c0 = W^T C from the fp32 weights w [Cout, Cin] and coef [Cout, 4]), x3d_pw_bwd (streams
    g and x only), x3d_pw_bwd_rc_finish (dw += from the moment sums).  x_stride = 2: the strided shortcut conv (x is the
    block input, g / dx live at the sampled pixels; epi = EPI_STORE, add = None).  Returns False (nothing launched) when the
    shape is not covered."""
    _chk(g, x, w, coef, dx, dw, add, tail_c, tail_r, tail_sums_c, tail_sums_r)
    n, cout, t, h, ww = g.shape
    cin = dx.shape[1]
    lib = hip.load()
    pe = int(lib.x3d_pw_bwd_rc_panel_elems(cout, cin))
    if pe == 0:
        return False
    panel = torch.empty(pe, dtype=g.dtype, device=g.device)
    c0 = torch.empty(cin, dtype=torch.float32, device=g.device)
    sums = torch.zeros(int(lib.x3d_pw_bwd_rc_sums_elems(cout, cin)), dtype=torch.float32, device=g.device)
    a = hip.PwBwdArgs(ptr(g), None, None, None, ptr(dx), epi, ptr(add), None, None, None, None, ptr(x), None, n, cin, cout,
                      t, h, ww, hip.dtype_code(g.dtype), ptr(tail_c), ptr(tail_r), ptr(tail_sums_c), ptr(tail_sums_r),
                      ptr(panel), ptr(c0), ptr(sums), x_stride, x.shape[3], x.shape[4])
    import ctypes as C
    if not lib.x3d_pw_bwd_supported(C.byref(a)):
        return False
    dt = hip.dtype_code(g.dtype)
    hip.call("x3d_pw_bwd_rc_prepare", ptr(w), ptr(coef), ptr(panel), ptr(c0), cout, cin, dt)
    hip.call_struct("x3d_pw_bwd", a)
    hip.call("x3d_pw_bwd_rc_finish", ptr(sums), ptr(w), ptr(coef), ptr(dw), cout, cin, dt)
    return True


def bn_bwd_finalize_rc(sums, count, mi, gamma, coef, dgamma, dbeta, dtype, prep=None, fin=None):
    """x3d_bn_bwd_finalize_rc: the BatchNorm-backward finalize + (prep = (w, panel, c0)) the recomputed-output panel of the conv
    in front of this BatchNorm + (fin = (sums, w, coef, dw)) the pending dW of an earlier x3d_pw_bwd, one launch."""
    w, panel, c0 = prep if prep is not None else (None, None, None)
    fs, fw, fc, fdw = fin if fin is not None else (None, None, None, None)
    _chk(sums, mi, gamma, coef, dgamma, dbeta, w, panel, c0, fs, fw, fc, fdw)
    hip.call("x3d_bn_bwd_finalize_rc", ptr(sums), float(count), ptr(mi), ptr(gamma), ptr(coef), ptr(dgamma), ptr(dbeta),
             gamma.numel(), ptr(w), ptr(panel), ptr(c0), 0 if w is None else w.shape[1], ptr(fs), ptr(fw), ptr(fc), ptr(fdw),
             0 if fw is None else fw.shape[0], 0 if fw is None else fw.shape[1], hip.dtype_code(dtype))


def pw_wgrad(g, yraw, coef, x, dw, in_ss=None, in_gate=None, in_act=ACT_NONE, stride=1, slab=False, coef_fold=None):
    """x: conv input [N,Cin,T,H,W] (input extents); g/yraw at the output points.  slab: through partial slabs +
    x3d_dw_slab_reduce (returns None, nothing launched, when the kernel behind the call has no slab form)."""
    _chk(g, yraw, coef, x, dw, in_ss, in_gate)
    n, cin, t, h, ww = x.shape
    cout = g.shape[1]
    a = hip.PwWgradArgs(ptr(g), ptr(yraw), ptr(coef), ptr(x), ptr(in_ss), ptr(in_gate), in_act, ptr(dw),
                        n, cin, cout, t, h, ww, stride, hip.dtype_code(x.dtype))
    if coef_fold is not None:
        a.coef_fold = hip.fold_address(coef_fold)
    if slab:
        import ctypes as C
        parts = int(hip.load().x3d_pw_wgrad_dw_parts(C.byref(a)))
        if parts <= 0:
            return None
        buf = torch.full((parts * cout * cin,), float("nan"), dtype=torch.float32, device=g.device)   # (every slab element must be written)
        a.dw_slab, a.dw_slab_parts = ptr(buf), parts
        hip.call_struct("x3d_pw_wgrad", a)
        dw_slab_reduce([(buf, dw, parts)])
        return True
    hip.call_struct("x3d_pw_wgrad", a)
    return True


# ---- depthwise ----------------------------------------------------------------------------------
def dw3d_fwd(x, w, stride, y=None, in_ss=None, in_act=ACT_NONE, stats=None, pool=None, in_bn=None):
    """in_bn: an ops.bn_fold(...) struct -- the prologue's BatchNorm finalize runs inside the kernel (in_ss unused)."""
    _chk(x, w, y, in_ss, stats, pool)
    n, c, t, h, ww = x.shape
    ho, wo = _out_hw(h, ww, stride)
    if y is None:
        y = torch.empty((n, c, t, ho, wo), dtype=x.dtype, device=x.device)
    st = _Stats(stats, c)
    a = hip.Dw3dFwdArgs(ptr(x), ptr(w), ptr(y), ptr(in_ss), in_act, ptr(st.arg()), ptr(pool), n, c, t, h,
                        ww, stride, hip.dtype_code(x.dtype), None if in_bn is None else ctypes.pointer(in_bn))
    hip.call_struct("x3d_dw3d_fwd", a)
    st.done()
    return y


def dw3d_bwd(dv, braw, coef_nc, araw, a_ss, w, ga, a_sums, dw, stride):
    _chk(dv, braw, coef_nc, araw, a_ss, w, ga, a_sums, dw)
    n, c, t, h, ww = araw.shape
    a = hip.Dw3dBwdArgs(ptr(dv), ptr(braw), ptr(coef_nc), ptr(araw), ptr(a_ss), ptr(w), ptr(ga),
                        ptr(a_sums), ptr(dw), n, c, t, h, ww, stride, hip.dtype_code(araw.dtype))
    hip.call_struct("x3d_dw3d_bwd", a)


# ---- squeeze-excite -----------------------------------------------------------------------------
def se_fwd(pool_sums, P, b_ss, w1, b1, w2, b2, gate, hidden):
    _chk(pool_sums, b_ss, w1, b1, w2, b2, gate, hidden)
    n, c = gate.shape
    hip.call("x3d_se_fwd", ptr(pool_sums), float(P), ptr(b_ss), ptr(w1), ptr(b1), ptr(w2), ptr(b2),
             ptr(gate), ptr(hidden), n, c, w1.shape[0])


def se_bnb_bwd(nc_sums, pool_sums, P, b_ss, b_mi, gamma_b, dgamma_b, dbeta_b, coef_nc, N, C,
               w1=None, b1=None, w2=None, b2=None, gate=None, hidden=None, dw1=None, db1=None,
               dw2=None, db2=None, scratch=None, reduce=()):
    """reduce: up to two (slab, dw, parts) weight-gradient slab jobs added up by extra workgroups of the launch."""
    _chk(nc_sums, pool_sums, b_ss, b_mi, gamma_b, dgamma_b, dbeta_b, coef_nc, w1, b1, w2, b2, gate,
         hidden, dw1, db1, dw2, db2, scratch)
    a = hip.SeBnbBwdArgs(ptr(nc_sums), ptr(pool_sums), float(P), ptr(b_ss), ptr(b_mi), ptr(gamma_b),
                         ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(gate), ptr(hidden), ptr(dw1),
                         ptr(db1), ptr(dw2), ptr(db2), ptr(dgamma_b), ptr(dbeta_b), ptr(coef_nc),
                         ptr(scratch), N, C, 0 if w1 is None else w1.shape[0])
    for i, job in enumerate(dw_reduce_jobs(list(reduce))[:len(reduce)]):
        a.reduce[i] = job
    hip.call_struct("x3d_se_bnb_bwd", a)


# ---- residual tail / reductions -----------------------------------------------------------------
def tail_fwd(c_raw, c_ss, shortcut, r_ss, y):
    _chk(c_raw, c_ss, shortcut, r_ss, y)
    n, c = c_raw.shape[:2]
    hip.call("x3d_tail_fwd", ptr(c_raw), ptr(c_ss), ptr(shortcut), ptr(r_ss), ptr(y), n, c,
             c_raw[0, 0].numel(), hip.dtype_code(c_raw.dtype))
    return y


def tail_fwd_bn(c_raw, c_bn, shortcut, r_bn, y):
    """tail_fwd with the finalize of bn_c (and bn_r) folded in; c_bn / r_bn: ops.bn_fold(...) structs."""
    _chk(c_raw, shortcut, y)
    n, c = c_raw.shape[:2]
    hip.call("x3d_tail_fwd_bn", ptr(c_raw), ctypes.byref(c_bn), ptr(shortcut), None if r_bn is None else ctypes.byref(r_bn),
             ptr(y), n, c, c_raw[0, 0].numel(), hip.dtype_code(c_raw.dtype))
    return y


def tail_bwd(dy_g, y, c_raw, r_raw, sums_c, sums_r):
    _chk(dy_g, y, c_raw, r_raw, sums_c, sums_r)
    n, c = y.shape[:2]
    hip.call("x3d_tail_bwd", ptr(dy_g), ptr(y), ptr(c_raw), ptr(r_raw), ptr(sums_c), ptr(sums_r), n, c,
             y[0, 0].numel(), hip.dtype_code(y.dtype))


def relu_bn_bwd_reduce(dy, dpool, yraw, ss, g, sums):
    _chk(dy, dpool, yraw, ss, g, sums)
    n, c = yraw.shape[:2]
    hip.call("x3d_relu_bn_bwd_reduce", ptr(dy), ptr(dpool), ptr(yraw), ptr(ss), ptr(g), ptr(sums), n, c,
             yraw[0, 0].numel(), hip.dtype_code(yraw.dtype))


def pool_fwd(x_raw, ss, pooled):
    _chk(x_raw, ss, pooled)
    n, c = x_raw.shape[:2]
    hip.call("x3d_pool_fwd", ptr(x_raw), ptr(ss), ptr(pooled), n, c, x_raw[0, 0].numel(),
             hip.dtype_code(x_raw.dtype))
    return pooled


# ---- head ---------------------------------------------------------------------------------------
def dense_fwd(x, w, b, y, act=ACT_NONE, mask=None, mask_scale=1.0):
    _chk(x, w, b, y, mask)
    n, k = x.shape
    hip.call("x3d_dense_fwd", ptr(x), ptr(mask), float(mask_scale), ptr(w), ptr(b), ptr(y), act, n, k,
             w.shape[0])
    return y


def dense_bwd(dy, y, act, x, w, dx, dw, db, mask=None, mask_scale=1.0):
    _chk(dy, y, x, w, dx, dw, db, mask)
    n, k = x.shape
    hip.call("x3d_dense_bwd", ptr(dy), ptr(y), act, ptr(x), ptr(mask), float(mask_scale), ptr(w),
             ptr(dx), ptr(dw), ptr(db), n, k, w.shape[0])


def softmax_xent(logits, labels, probs, loss_rows=None, dlogits=None, grad_scale=1.0):
    _chk(logits, labels, probs, loss_rows, dlogits)
    n, m = logits.shape
    hip.call("x3d_softmax_xent", ptr(logits), ptr(labels), ptr(probs), ptr(loss_rows), ptr(dlogits),
             float(grad_scale), n, m)
    return probs


def view_mean(probs, out, views):
    _chk(probs, out)
    hip.call("x3d_view_mean", ptr(probs), ptr(out), out.shape[0], views, probs.shape[1])
    return out


def sgd_nesterov(w, v, g, l2_mask, lr, momentum, weight_decay, grad_scale=1.0):
    _chk(w, v, g, l2_mask)
    hip.call("x3d_sgd_nesterov", ptr(w), ptr(v), ptr(g), ptr(l2_mask), float(lr), float(momentum),
             float(weight_decay), float(grad_scale), w.numel())


def l2_sumsq(w, l2_mask, out):
    _chk(w, l2_mask, out)
    hip.call("x3d_l2_sumsq", ptr(w), ptr(l2_mask), ptr(out), w.numel())


def nthwc_to_ncthw(src, dst):
    """src [N,T,H,W,C] -> dst [N,C,T,H,W] (with dtype conversion)."""
    _chk(src, dst)
    n, t, h, w, c = src.shape
    hip.call("x3d_nthwc_to_ncthw", ptr(src), hip.dtype_code(src.dtype), ptr(dst),
             hip.dtype_code(dst.dtype), n, c, t * h * w)
    return dst


def subsample2(x, out=None):
    """Even-pixel copy x[..., ::2, ::2] of an NCTHW tensor (x3d_subsample2): what a stride-(1,2,2) shortcut conv samples."""
    _chk(x, out)
    n, c, t, h, w = x.shape
    if out is None:
        out = torch.empty((n, c, t, (h + 1) // 2, (w + 1) // 2), dtype=x.dtype, device=x.device)
    hip.call("x3d_subsample2", ptr(x), ptr(out), n * c * t, h, w, hip.dtype_code(x.dtype))
    return out
