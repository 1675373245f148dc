"""Kernel-level parity: every C-ABI entry point against a CPU fp64 restatement of the same op
(oracle/x3d_oracle.py for the conv primitives, torch autograd for the gradients) on shapes small
enough for the CPU to finish in seconds.  Shapes cover stride 1/2, even/odd extents (TF-SAME 0/1 and
1/1 pads, incl. X3D-L's 39 -> 20), channel counts that are not multiples of the MFMA tile, and point
counts that are not multiples of the vector width.
"""
import pytest
import torch
import torch.nn.functional as F

from tests import shapes as S
from tests.util import report, rnd, round_to, tie_slack, tol_gemm, tol_store

pytestmark = pytest.mark.gpu

DTYPES = S.DTYPES            # fp32, bf16, fp16 activation storage
HALF = S.HALF_DTYPES


def _ops():
    from x3d_tf_amd import ops
    return ops


def hip_lib():
    from x3d_tf_amd import hip
    return hip.load()


def _oracle():
    from oracle import x3d_oracle as O
    return O


def _gen(seed=0):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def _stol(dtype):
    """tolerance of the epilogue sums.  They are taken from the fp32 values BEFORE the store rounds them; for bf16
    storage they equal the sums of the stored tensor only to ~2^-9/sqrt(count) (unbiased rounding)."""
    return 1e-5 if dtype == torch.float32 else (3e-3 if dtype == torch.bfloat16 else 5e-4)


def _dw_on_matrix_cores(st):
    """does x3d_dw3d_fwd / _bwd dispatch this launch to a matrix-core kernel (dw_mx.hip)?  Those round their operands
    to the storage type (the vector kernels multiply in fp32), so the reference is the fp64 convolution of the rounded
    operands, at the pointwise kernels' tolerance."""
    from x3d_tf_amd import hip
    return "_mx" in hip.dw3d_kernel_name(st)


def _stats_ref(y, dtype):
    yr = y.to(dtype).double()
    return torch.stack([yr.sum((0, 2, 3, 4)), (yr * yr).sum((0, 2, 3, 4))], 1)


def _affine(x, ss, gate=None, act=0):
    u = x * ss[:, 0].view(1, -1, 1, 1, 1) + ss[:, 1].view(1, -1, 1, 1, 1)
    if gate is not None:
        u = u * gate[:, :, None, None, None]
    if act == 1:
        u = F.relu(u)
    elif act == 2:
        u = u * torch.sigmoid(u)
    return u


# --------------------------------------------------------------------------------------------------
def _panels(ops, wt, dtype, gpu):
    (fp, dp), = ops.pw_pack_weights([wt.to(gpu)], dtype=dtype)
    return fp, dp


@pytest.mark.parametrize("panel", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", S.PW_FWD + S.PW_FWD_XL)
def test_pw_fwd(gpu, dtype, shape, panel):
    """x3d_pw_fwd against the oracle's pointwise conv.  panel=True passes the packed weight panel, as model.py always does
    for 16-bit storage: that is the production dispatch (weights-stationary / weights-streamed / resident-panel kernels)."""
    if panel and dtype == torch.float32:
        pytest.skip("weight panels exist for the 16-bit storage types only")
    ops, O = _ops(), _oracle()
    n, cin, cout, t, h, w, stride, pro = shape
    g = _gen(1)
    x, xd = rnd((n, cin, t, h, w), dtype, g)
    wt = torch.randn((cout, cin), generator=g) * 0.2
    ss = gate = None
    act = 0
    xin = xd
    if pro:
        ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g)], 1)
        act = 2 if pro == "swish" else 1
        if pro == "swish":
            gate = torch.rand((n, cin), generator=g)
        xin = _affine(x.float(), ss, gate, act)       # the kernel's prologue runs in fp32 ...
    # ... and its output and the weights are rounded to the storage type for the matrix cores: the reference GEMM takes
    # the operands as the matrix cores see them
    ref = O.pointwise(round_to(xin, dtype), round_to(wt, dtype), stride)
    stats = torch.zeros((cout, 2), dtype=torch.float64, device=gpu)
    fp = _panels(ops, wt, dtype, gpu)[0] if panel else None
    y = ops.pw_fwd(x.to(gpu), wt.to(gpu), stats=stats, in_ss=None if ss is None else ss.to(gpu),
                   in_gate=None if gate is None else gate.to(gpu), in_act=act, stride=stride, w_panel=fp)
    torch.cuda.synchronize()
    rt, at = tol_gemm(dtype)
    scale = ref.abs().max().item()
    slack = 0.0
    if pro and dtype != torch.float32 and stride == 1:   # an operand on a rounding tie may land on the other neighbour (util.tie_slack)
        slack = tie_slack(xin, dtype, round_to(wt, dtype).abs())
    report("y", y, ref, rt, at * scale + slack)
    # statistics describe the tensor as stored
    sref = _stats_ref(y.float().cpu(), dtype)
    report("stats", stats, sref, _stol(dtype), _stol(dtype) * max(1.0, sref.abs().max().item()))


@pytest.mark.parametrize("panel", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", S.PW_FWD_TAIL)
def test_pw_fwd_tail(gpu, dtype, shape, panel):
    """x3d_pw_fwd with the residual tail of the block below folded into its prologue (in_add / in_store): x is that block's
    raw `c` output, the conv input y = relu(bn_c(x) + shortcut) (reference model.py:381-392) is built on load, STORED (it is
    the block's output: the shortcut conv, the next tail and the backward pass read it) and multiplied.  y against the fp64
    formula at the storage tolerance, bit-identical to what x3d_tail_fwd stores up to one rounding of the fused affine; the
    conv against the oracle on the stored y."""
    if panel and dtype == torch.float32:
        pytest.skip("weight panels exist for the 16-bit storage types only")
    ops, O = _ops(), _oracle()
    n, cin, cout, t, h, w, stride, pro = shape
    g = _gen(19)
    c, cd = rnd((n, cin, t, h, w), dtype, g)
    sc_, scd = rnd((n, cin, t, h, w), dtype, g)
    wt = torch.randn((cout, cin), generator=g) * 0.2
    ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g)], 1)
    ssr = None
    short = scd
    if pro == "tail_conv":
        ssr = torch.stack([1 + 0.3 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g)], 1)
        short = _affine(scd, ssr.double())
    if pro == "tail1":          # no Add: the stem's BatchNorm + ReLU
        sc_, short = None, 0.0
    y_ref = F.relu(_affine(cd, ss.double()) + short)
    dev = lambda v: None if v is None else v.to(gpu)
    ystore = torch.full((n, cin, t, h, w), 7.0, dtype=dtype, device=gpu)
    stats = torch.zeros((cout, 2), dtype=torch.float64, device=gpu)
    fp = _panels(ops, wt, dtype, gpu)[0] if panel else None
    out = ops.pw_fwd(dev(c), dev(wt), stats=stats, in_ss=dev(ss), in_act=1, w_panel=fp, in_add=dev(sc_), in_add_ss=dev(ssr),
                     in_store=ystore)
    torch.cuda.synchronize()
    rs, as_ = tol_store(dtype)
    report("y (stored conv input)", ystore, y_ref, rs, as_ * y_ref.abs().max().item())
    # the separate pass stores the same tensor (the two evaluate the affine sum in a different association: one ulp at most)
    y2 = torch.empty_like(ystore)
    ops.tail_fwd(dev(c), dev(ss), dev(sc_), dev(ssr), y2)
    torch.cuda.synchronize()
    report("y vs x3d_tail_fwd", ystore, y2.float().cpu(), 2 * rs, as_ * y_ref.abs().max().item())
    ref = O.pointwise(ystore.float().cpu().double(), round_to(wt, dtype), 1)      # the GEMM operand is the stored y
    rt, at = tol_gemm(dtype)
    report("conv", out, ref, rt, at * ref.abs().max().item())
    sref = _stats_ref(out.float().cpu(), dtype)
    report("stats", stats, sref, _stol(dtype), _stol(dtype) * max(1.0, sref.abs().max().item()))


@pytest.mark.parametrize("panel", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", S.PW_FWD_INFER)
def test_pw_fwd_infer(gpu, dtype, shape, panel):
    """x3d_pw_fwd with the inference epilogue (out_scale_shift): y = act(s_o * conv(f(x)) + t_o [+ s_r * add + t_r]) -- the
    BatchNorm after the conv folded from the moving statistics and the residual Add + ReLU (reference model.py:300-303,
    368-371, 381-392 at training=False) -- against the oracle's pointwise conv + the same affine arithmetic in fp64."""
    if panel and dtype == torch.float32:
        pytest.skip("weight panels exist for the 16-bit storage types only")
    ops, O = _ops(), _oracle()
    n, cin, cout, t, h, w, pro, res, oact = shape
    g = _gen(17)
    x, xd = rnd((n, cin, t, h, w), dtype, g)
    wt = torch.randn((cout, cin), generator=g) * 0.2
    ss = gate = None
    act = 0
    xin = xd
    if pro:
        ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g)], 1)
        act = 2 if pro == "swish" else 1
        if pro == "swish":
            gate = torch.rand((n, cin), generator=g)
        xin = _affine(x.float(), ss, gate, act)
    oss = torch.stack([1 + 0.3 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)], 1)
    ref = _affine(O.pointwise(round_to(xin, dtype), round_to(wt, dtype), 1), oss.double())
    add = ass = None
    if res:
        add, addd = rnd((n, cout, t, h, w), dtype, g)
        if res == "conv":
            ass = torch.stack([1 + 0.3 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)], 1)
            addd = _affine(addd, ass.double())
        ref = ref + addd
    if oact == "relu":
        ref = F.relu(ref)
    fp = _panels(ops, wt, dtype, gpu)[0] if panel else None
    dev = lambda v: None if v is None else v.to(gpu)
    y = ops.pw_fwd(dev(x), dev(wt), in_ss=dev(ss), in_gate=dev(gate), in_act=act, w_panel=fp, out_ss=dev(oss),
                   out_add=dev(add), out_add_ss=dev(ass), out_act=1 if oact == "relu" else 0)
    torch.cuda.synchronize()
    rt, at = tol_gemm(dtype)
    slack = 0.0
    if pro and dtype != torch.float32:     # a prologue value on a rounding tie (util.tie_slack), through the output scale
        slack = tie_slack(xin, dtype, round_to(wt, dtype).abs()) * oss[:, 0].abs().double().view(1, -1, 1, 1, 1)
    report("y", y, ref, rt, at * max(ref.abs().max().item(), 1.0) + slack)
    # the training form of the same launch still refuses the epilogue operands it cannot honour
    with pytest.raises(Exception):
        ops.pw_fwd(dev(x), dev(wt), stats=torch.zeros((cout, 2), dtype=torch.float64, device=gpu), out_ss=dev(oss))


def _dyraw(coef, gd, yd, dtype=None):
    """dYraw = A*g + B*yraw + C (BatchNorm backward folded into the load).  dtype given: evaluated in fp32 like the kernel's
    prologue and rounded to the storage type (the GEMM operand as the matrix cores see it), returned as fp64."""
    if dtype is not None:
        c = coef.float()
        v = c[:, 0].view(1, -1, 1, 1, 1) * gd.float() + c[:, 1].view(1, -1, 1, 1, 1) * yd.float() + c[:, 2].view(1, -1, 1, 1, 1)
        return round_to(v, dtype)
    c = coef.double()
    return c[:, 0].view(1, -1, 1, 1, 1) * gd + c[:, 1].view(1, -1, 1, 1, 1) * yd + c[:, 2].view(1, -1, 1, 1, 1)


def _dyraw_slack(coef, gd, yd, dtype, wt):
    """tie slack (util.tie_slack) of a data gradient dx = Wr^T dYraw whose operand dYraw = A g + B yraw + C is evaluated in
    fp32 and rounded to the 16-bit storage type: where the fp32 value sits on a rounding tie the kernel's fused multiply-adds
    may land on the other neighbour than torch's association does.  [N, Cin, T, H, W]; zero almost everywhere."""
    if dtype == torch.float32:
        return 0.0
    c = coef.float()
    v = c[:, 0].view(1, -1, 1, 1, 1) * gd.float() + c[:, 1].view(1, -1, 1, 1, 1) * yd.float() + c[:, 2].view(1, -1, 1, 1, 1)
    return tie_slack(v, dtype, round_to(wt, dtype).abs().t().contiguous())


def _wtol(dtype):
    """weight gradients against an fp64 GEMM over operands rounded like the kernel's: fp32 output, so what is left is the
    fp32 accumulation order (partial tiles, atomics) and the rare operand on a rounding boundary (tol_gemm)."""
    return 2e-4 if dtype == torch.float32 else (1e-3 if dtype == torch.bfloat16 else 4e-4)


@pytest.mark.parametrize("slab", [False, True])
@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("shape", S.PW_BWD)
def test_pw_bwd_oracle(gpu, dtype, shape, slab):
    """x3d_pw_bwd (fused data + weight gradient, one pass over dY) against an fp64 restatement of both gradients --
    directly, not through the unfused kernels (test_pw_bwd_fused below keeps the bit-for-bit comparison with those).
    slab: the weight gradient through per-workgroup partial slabs + x3d_dw_slab_reduce (x3d_hip.h dw_slab) -- the form the
    plans use where the kernel behind the call has it (the persistent weights-stationary kernels)."""
    ops = _ops()
    n, cin, cout, t, h, w, epi = shape
    if slab and not hip_lib().x3d_pw_bwd_dw_parts(S.pw_bwd_struct(shape, dtype)):
        pytest.skip("no slab form behind this call")
    g_ = _gen(31)
    gy, gyd = rnd((n, cout, t, h, w), dtype, g_)
    yraw, yrd = rnd((n, cout, t, h, w), dtype, g_)
    coef = torch.randn((cout, 4), generator=g_) * 0.5
    wt = torch.randn((cout, cin), generator=g_) * 0.2
    dy = _dyraw(coef, gyd, yrd, dtype)
    dx_ref = torch.einsum("oc,nothw->ncthw", round_to(wt, dtype), dy)
    dp = _panels(ops, wt, dtype, gpu)[1]
    dx = torch.empty((n, cin, t, h, w), dtype=dtype, device=gpu)
    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)     # += semantics
    dev = lambda v: v.to(gpu)
    if epi == "swish_bwd":
        braw, bd = rnd((n, cin, t, h, w), dtype, g_)
        bss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1)
        gate = torch.rand((n, cin), generator=g_)
        v = _affine(bd, bss.double(), gate.double(), 0)
        sg = torch.sigmoid(v)
        dx_ref = dx_ref * (sg * (1 + v * (1 - sg)))
        xin = round_to(_affine(braw.float(), bss, gate, 2), dtype)       # the conv input: swish(gate * bn_b(braw))
        ncs = torch.zeros((n, cin, 2), dtype=torch.float64, device=gpu)
        ok = ops.pw_bwd(dev(gy), dev(yraw), dev(coef), dp, dx, dw, ops.EPI_SWISH_BWD, braw=dev(braw), b_ss=dev(bss),
                        gate=dev(gate), nc_sums=ncs, slab=slab)
    else:
        x, xd = rnd((n, cin, t, h, w), dtype, g_)
        xin = xd
        if epi == "add":
            add, addd = rnd((n, cin, t, h, w), dtype, g_)
            dx_ref = dx_ref + addd
            e = ops.EPI_ADD
        else:
            add, addd = rnd((n, cin, t, (h + 1) // 2, (w + 1) // 2), dtype, g_)
            up = torch.zeros_like(dx_ref)
            up[:, :, :, ::2, ::2] = addd
            dx_ref = dx_ref + up
            e = ops.EPI_ADD_STRIDED
        ok = ops.pw_bwd(dev(gy), dev(yraw), dev(coef), dp, dx, dw, e, x=dev(x), add=dev(add), slab=slab)
    torch.cuda.synchronize()
    assert ok, "fused kernel should cover this shape"
    rt, at = tol_gemm(dtype)
    # (swish' <= 1.1: the epilogue cannot enlarge an operand flip by more)
    report("dx", dx, dx_ref, rt, at * dx_ref.abs().max().item() + 1.1 * _dyraw_slack(coef, gyd, yrd, dtype, wt))
    dw_ref = torch.einsum("nothw,ncthw->oc", dy, xin)
    tol = _wtol(dtype)
    report("dw", dw, dw_ref + 0.5, tol, tol * dw_ref.abs().max().item())
    if epi == "swish_bwd":
        dvs = dx.float().cpu().double()
        sref = torch.stack([dvs.sum((2, 3, 4)), (dvs * bd).sum((2, 3, 4))], -1)
        report("nc_sums", ncs, sref, 10 * _stol(dtype), 10 * _stol(dtype) * max(1.0, sref.abs().max().item()))


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("shape", S.PW_BWD_RC)
def test_pw_bwd_rc(gpu, dtype, shape):
    """x3d_pw_bwd with rc_panel (pw_bwd_rc.hip): the `a`-conv backward that never reads the conv's raw output.  Reference:
    the fp64 restatement of what the plain form computes -- y = Wr x (Wr: weights rounded to the storage type, as the forward
    GEMM multiplies them), dY = A g + B y + C, dx = Wr^T dY (+ add, + folded tail), dw = dY x^T -- so the algebraic fold
    dx = [Wr^T A | Wr^T B Wr] [g ; x] + Wr^T C,  dw = A (g x^T) + B Wr (x x^T) + C (sum x)^T is checked against the definition,
    not against itself: dw directly (tol of the weight gradients), dx in two steps -- the device against the fold with its
    two panel operands rounded to the storage type (tol_gemm, as every matrix-core test rounds its operands), and that
    rounded fold against the definition (the operand-rounding bound)."""
    ops = _ops()
    n, cin, cout, t, h, w, epi, tail = shape
    g_ = _gen(37)
    gy, gyd = rnd((n, cout, t, h, w), dtype, g_)
    coef = torch.randn((cout, 4), generator=g_) * 0.5
    wt = torch.randn((cout, cin), generator=g_) * 0.2
    x, xd = rnd((n, cin, t, h, w), dtype, g_)
    if tail:
        x = torch.relu(x)          # = output of the block below: about half of it positive
        xd = x.double()
    wr = round_to(wt, dtype)
    yd = torch.einsum("oc,ncthw->nothw", wr, xd)
    c = coef.double()
    dy = c[:, 0].view(1, -1, 1, 1, 1) * gyd + c[:, 1].view(1, -1, 1, 1, 1) * yd + c[:, 2].view(1, -1, 1, 1, 1)
    dx_def = torch.einsum("oc,nothw->ncthw", wr, dy)                  # the definition: W^T (A g + B (W x) + C)
    # ... and the same map with the two matrix-core operands of the kernel rounded to the storage type, as every
    # pointwise test rounds its operands (tol_gemm): W1 = Wr^T diag(A), M = Wr^T diag(B) Wr; c0 = Wr^T C stays fp32
    w1 = round_to((wr * c[:, 0:1]).float(), dtype)                     # [co][ci]
    mm = round_to(torch.einsum("oc,o,od->cd", wr, c[:, 1], wr).float(), dtype)
    dx_ref = (torch.einsum("oc,nothw->ncthw", w1, gyd) + torch.einsum("cd,ndthw->ncthw", mm, xd) +
              (wr * c[:, 2:3]).sum(0).view(1, -1, 1, 1, 1))
    # the algebraic fold IS the definition up to that operand rounding (2^-9 relative per panel entry for bf16)
    fold_err = (dx_ref - dx_def).abs().max().item() / dx_def.abs().max().item()
    assert fold_err < (2e-2 if dtype == torch.bfloat16 else 3e-3), fold_err
    if epi == "add":
        add, addd = rnd((n, cin, t, h, w), dtype, g_)
        dx_ref = dx_ref + addd
        e = ops.EPI_ADD
    else:
        add, addd = rnd((n, cin, t, (h + 1) // 2, (w + 1) // 2), dtype, g_)
        up = torch.zeros_like(dx_ref)
        up[:, :, :, ::2, ::2] = addd
        dx_ref = dx_ref + up
        e = ops.EPI_ADD_STRIDED
    dev = lambda v: None if v is None else v.to(gpu)
    craw = rraw = sc = sr = None
    if tail:
        craw, crd = rnd((n, cin, t, h, w), dtype, g_)
        sc = torch.zeros((cin, 2), dtype=torch.float64, device=gpu)
        if tail == 2:
            rraw, rrd = rnd((n, cin, t, h, w), dtype, g_)
            sr = torch.zeros((cin, 2), dtype=torch.float64, device=gpu)
    dx = torch.empty((n, cin, t, h, w), dtype=dtype, device=gpu)
    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)     # += semantics
    ok = ops.pw_bwd_rc(dev(gy), dev(x), dev(wt), dev(coef), dx, dw, e, dev(add), tail_c=dev(craw), tail_r=dev(rraw),
                       tail_sums_c=sc, tail_sums_r=sr)
    torch.cuda.synchronize()
    assert ok, "the recomputed-output form should cover this shape"
    rt, at = tol_gemm(dtype)
    scale = dx_ref.abs().max().item()
    if tail:
        # the mask is [x > 0] -- exact; the masked entries are exact zeros
        keep = xd > 0
        assert (dx.float().cpu()[~keep] == 0).all()
        dx_ref = dx_ref * keep
    report("dx", dx, dx_ref, rt, at * scale)
    dw_ref = torch.einsum("nothw,ncthw->oc", dy, xd)
    tol = _wtol(dtype)
    report("dw", dw, dw_ref + 0.5, tol, tol * dw_ref.abs().max().item())
    if tail:
        dxs = dx.float().cpu().double()      # the sums describe dx as stored
        sref = torch.stack([dxs.sum((0, 2, 3, 4)), (dxs * crd).sum((0, 2, 3, 4))], 1)
        report("tail_sums_c", sc, sref, 10 * _stol(dtype), 10 * _stol(dtype) * max(1.0, sref.abs().max().item()))
        if tail == 2:
            sref = torch.stack([dxs.sum((0, 2, 3, 4)), (dxs * rrd).sum((0, 2, 3, 4))], 1)
            report("tail_sums_r", sr, sref, 10 * _stol(dtype), 10 * _stol(dtype) * max(1.0, sref.abs().max().item()))


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("shape", S.PW_BWD_RC_STRIDED)
def test_pw_bwd_rc_strided(gpu, dtype, shape):
    """The strided shortcut conv's backward (reference model.py:360-371: residual 1x1x1 stride (1, 2, 2) -> bn_r) in the
    recomputed-output form: one launch streams g and the even pixels of the block input, no raw shortcut output.  Reference as
    in test_pw_bwd_rc, on xs = x[..., ::2, ::2]."""
    ops = _ops()
    n, cin, cout, t, xh, xw = shape
    ho, wo = (xh + 1) // 2, (xw + 1) // 2
    g_ = _gen(38)
    gy, gyd = rnd((n, cout, t, ho, wo), dtype, g_)
    coef = torch.randn((cout, 4), generator=g_) * 0.5
    wt = torch.randn((cout, cin), generator=g_) * 0.2
    x, xd = rnd((n, cin, t, xh, xw), dtype, g_)
    xs = xd[:, :, :, ::2, ::2]
    wr = round_to(wt, dtype)
    c = coef.double()
    yd = torch.einsum("oc,ncthw->nothw", wr, xs)
    dy = c[:, 0].view(1, -1, 1, 1, 1) * gyd + c[:, 1].view(1, -1, 1, 1, 1) * yd + c[:, 2].view(1, -1, 1, 1, 1)
    dx_def = torch.einsum("oc,nothw->ncthw", wr, dy)
    w1 = round_to((wr * c[:, 0:1]).float(), dtype)
    mm = round_to(torch.einsum("oc,o,od->cd", wr, c[:, 1], wr).float(), dtype)
    dx_ref = (torch.einsum("oc,nothw->ncthw", w1, gyd) + torch.einsum("cd,ndthw->ncthw", mm, xs) +
              (wr * c[:, 2:3]).sum(0).view(1, -1, 1, 1, 1))
    fold_err = (dx_ref - dx_def).abs().max().item() / dx_def.abs().max().item()
    assert fold_err < (2e-2 if dtype == torch.bfloat16 else 3e-3), fold_err
    dev = lambda v: v.to(gpu)
    dx = torch.empty((n, cin, t, ho, wo), dtype=dtype, device=gpu)
    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)
    ok = ops.pw_bwd_rc(dev(gy), dev(x), dev(wt), dev(coef), dx, dw, ops.EPI_STORE, None, x_stride=2)
    torch.cuda.synchronize()
    assert ok, "the strided recomputed-output form should cover this shape"
    rt, at = tol_gemm(dtype)
    report("dx", dx, dx_ref, rt, at * dx_ref.abs().max().item())
    dw_ref = torch.einsum("nothw,ncthw->oc", dy, xs)
    tol = _wtol(dtype)
    report("dw", dw, dw_ref + 0.5, tol, tol * dw_ref.abs().max().item())
    # round 6, the plans' path: the same gradients from the DENSE store form on the even-pixel copy of x (x3d_subsample2)
    xc = ops.subsample2(dev(x))
    assert torch.equal(xc, dev(x)[..., ::2, ::2].contiguous())
    dx2 = torch.empty_like(dx)
    dw2 = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)
    ok2 = ops.pw_bwd_rc(dev(gy), xc, dev(wt), dev(coef), dx2, dw2, ops.EPI_STORE, None)
    torch.cuda.synchronize()
    if ok2:      # (P % 8 != 0 on the compact plane: the dense form declines, the plan keeps the strided one)
        report("dx (compact)", dx2, dx_ref, rt, at * dx_ref.abs().max().item())
        report("dw (compact)", dw2, dw_ref + 0.5, tol, tol * dw_ref.abs().max().item())


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("cout,cin,fco,fci", [(54, 24, 108, 24), (108, 48, 54, 24), (72, 32, 72, 32), (20, 8, 127, 24), (216, 48, 108, 48), (54, 24, 216, 48)])
def test_bn_bwd_finalize_rc_equals_the_three_launches(gpu, dtype, cout, cin, fco, fci):
    """x3d_bn_bwd_finalize_rc (finalize + panel of this layer + dW of an earlier layer in one launch) writes the same bits as
    x3d_bn_bwd_finalize, x3d_pw_bwd_rc_prepare and x3d_pw_bwd_rc_finish one after the other -- with and without each job."""
    ops = _ops()
    from x3d_tf_amd import hip
    lib = hip.load()
    g_ = _gen(39)
    dev = lambda v: v.to(gpu)
    sums = dev(torch.randn((cout, 2), generator=g_, dtype=torch.float64) * 50)
    mi = dev(torch.stack([torch.randn(cout, generator=g_), 0.5 + torch.rand(cout, generator=g_)], 1))
    gamma = dev(1 + 0.2 * torch.randn(cout, generator=g_))
    w = dev(torch.randn((cout, cin), generator=g_) * 0.2)
    fw = dev(torch.randn((fco, fci), generator=g_) * 0.2)
    fcoef = dev(torch.randn((fco, 4), generator=g_) * 0.5)
    fsums = dev(torch.randn((fco + 1 + fci) * fci, generator=g_) * 30)
    count = 12345.0
    pe = int(lib.x3d_pw_bwd_rc_panel_elems(cout, cin))
    assert pe > 0

    def fresh():
        return (torch.zeros((cout, 4), device=gpu), torch.full((cout,), 0.25, device=gpu), torch.full((cout,), -0.5, device=gpu),
                torch.zeros(pe, dtype=dtype, device=gpu), torch.zeros(cin, device=gpu), torch.full((fco, fci), 0.5, device=gpu))
    coef0, dg0, db0, pan0, c00, dw0 = fresh()
    ops.bn_bwd_finalize(sums, count, mi, gamma, coef0, dg0, db0)
    hip.call("x3d_pw_bwd_rc_prepare", w.data_ptr(), coef0.data_ptr(), pan0.data_ptr(), c00.data_ptr(), cout, cin, hip.dtype_code(dtype))
    hip.call("x3d_pw_bwd_rc_finish", fsums.data_ptr(), fw.data_ptr(), fcoef.data_ptr(), dw0.data_ptr(), fco, fci, hip.dtype_code(dtype))
    for with_prep in (True, False):
        for with_fin in (True, False):
            coef1, dg1, db1, pan1, c01, dw1 = fresh()
            ops.bn_bwd_finalize_rc(sums, count, mi, gamma, coef1, dg1, db1, dtype, prep=(w, pan1, c01) if with_prep else None,
                                   fin=(fsums, fw, fcoef, dw1) if with_fin else None)
            torch.cuda.synchronize()
            assert torch.equal(coef0, coef1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
            if with_prep:
                assert torch.equal(pan0, pan1) and torch.equal(c00, c01)
            if with_fin:
                assert torch.equal(dw0, dw1)
            else:
                assert float((dw1 - 0.5).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("shape", S.PW_BWD_TAIL)
def test_pw_bwd_tail(gpu, dtype, shape):
    """x3d_pw_bwd with the residual-tail backward of the block below folded into its epilogue (tail_c / tail_r): the conv
    input x is that block's output y, so dx = [x > 0] * (W^T dY + add) -- what x3d_tail_bwd would have made of the unmasked
    dx in a separate pass -- with the BN_c / BN_r backward sums (sum dx, sum dx * c_raw | r_raw).  Against an fp64
    restatement; dW is unaffected by the fold."""
    ops = _ops()
    n, cin, cout, t, h, w, epi, tail = shape
    g_ = _gen(33)
    gy, gyd = rnd((n, cout, t, h, w), dtype, g_)
    yraw, yrd = rnd((n, cout, t, h, w), dtype, g_)
    coef = torch.randn((cout, 4), generator=g_) * 0.5
    wt = torch.randn((cout, cin), generator=g_) * 0.2
    dy = _dyraw(coef, gyd, yrd, dtype)
    dx_ref = torch.einsum("oc,nothw->ncthw", round_to(wt, dtype), dy)
    dp = _panels(ops, wt, dtype, gpu)[1]
    x, xd = rnd((n, cin, t, h, w), dtype, g_)           # = relu output of the block below: about half of it positive
    x = torch.relu(x)
    xd = x.double()
    if epi == "add":
        add, addd = rnd((n, cin, t, h, w), dtype, g_)
        dx_ref = dx_ref + addd
        e = ops.EPI_ADD
    else:
        add, addd = rnd((n, cin, t, (h + 1) // 2, (w + 1) // 2), dtype, g_)
        up = torch.zeros_like(dx_ref)
        up[:, :, :, ::2, ::2] = addd
        dx_ref = dx_ref + up
        e = ops.EPI_ADD_STRIDED
    mask = xd > 0
    g_ref = dx_ref * mask
    tc, tcd = rnd((n, cin, t, h, w), dtype, g_)
    tr, trd = rnd((n, cin, t, h, w), dtype, g_)
    dx = torch.empty((n, cin, t, h, w), dtype=dtype, device=gpu)
    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)
    sc = torch.full((cin, 2), 0.25, dtype=torch.float64, device=gpu)     # += semantics
    sr = torch.full((cin, 2), 0.25, dtype=torch.float64, device=gpu)
    dev = lambda v: v.to(gpu)
    ok = ops.pw_bwd(dev(gy), dev(yraw), dev(coef), dp, dx, dw, e, x=dev(x), add=dev(add), tail_c=dev(tc),
                    tail_r=dev(tr) if tail == 2 else None, tail_sums_c=sc, tail_sums_r=sr if tail == 2 else None)
    torch.cuda.synchronize()
    assert ok, "the fused kernel with the tail epilogue should cover this shape"
    rt, at = tol_gemm(dtype)
    report("dx (masked)", dx, g_ref, rt, at * dx_ref.abs().max().item())
    assert not bool((dx.float().cpu()[~mask] != 0).any()), "gradient leaked through a closed ReLU"
    dw_ref = torch.einsum("nothw,ncthw->oc", dy, xd)
    report("dw", dw, dw_ref + 0.5, _wtol(dtype), _wtol(dtype) * dw_ref.abs().max().item())
    gs = dx.float().cpu().double()                       # the sums describe the gradient as stored
    st = 10 * _stol(dtype)
    ref_c = torch.stack([gs.sum((0, 2, 3, 4)), (gs * tcd).sum((0, 2, 3, 4))], 1) + 0.25
    report("tail_sums_c", sc, ref_c, st, st * max(1.0, ref_c.abs().max().item()))
    if tail == 2:
        ref_r = torch.stack([gs.sum((0, 2, 3, 4)), (gs * trd).sum((0, 2, 3, 4))], 1) + 0.25
        report("tail_sums_r", sr, ref_r, st, st * max(1.0, ref_r.abs().max().item()))
    else:
        assert float((sr - 0.25).abs().max()) == 0.0
    # the weight gradient through partial slabs (x3d_hip.h dw_slab; what the plans record where the kernel has the form)
    if hip_lib().x3d_pw_bwd_dw_parts(S.pw_bwd_struct(shape, dtype)):
        dx3 = torch.empty_like(dx)
        dw3 = torch.full_like(dw, 0.5)
        sc3 = torch.full_like(sc, 0.25)
        assert ops.pw_bwd(dev(gy), dev(yraw), dev(coef), dp, dx3, dw3, e, x=dev(x), add=dev(add), tail_c=dev(tc), tail_sums_c=sc3, slab=True)
        torch.cuda.synchronize()
        assert torch.equal(dx3, dx)
        report("dw (slabs)", dw3, dw_ref + 0.5, _wtol(dtype), _wtol(dtype) * dw_ref.abs().max().item())
    # the same launch without the fold followed by x3d_tail_bwd gives the same masked gradient bit for bit
    dx2 = torch.empty_like(dx)
    dw2 = torch.zeros_like(dw)
    ops.pw_bwd(dev(gy), dev(yraw), dev(coef), dp, dx2, dw2, e, x=dev(x), add=dev(add))
    s2c = torch.zeros((cin, 2), dtype=torch.float64, device=gpu)
    s2r = torch.zeros((cin, 2), dtype=torch.float64, device=gpu)
    ops.tail_bwd(dx2, dev(x), dev(tc), dev(tr) if tail == 2 else None, s2c, s2r if tail == 2 else None)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
    report("sums vs x3d_tail_bwd", sc - 0.25, s2c.cpu(), 1e-6, 1e-6 * max(1.0, ref_c.abs().max().item()))   # both sum the STORED gradient


# the BatchNorm-backward finalize derived by its consumers (include/x3d_hip.h x3d_bn_bwd_fold): (N, Cin, Cout, T, H, W, kind)
COEF_FOLD = [
    (2, 216, 96, 2, 14, 14, "c"), (2, 432, 192, 8, 7, 7, "c"),           # pw_bwd_wst.hip: stage 4, stage 5 (two slices)
    (2, 96, 216, 2, 14, 14, "a"), (2, 96, 216, 2, 14, 14, "a_tail"),      # pw_bwd_wsta.hip
    (2, 192, 432, 8, 7, 7, "pair"), (2, 432, 192, 8, 7, 7, "pair_c"),     # x3d_pw_wgrad (12-tile groups) + x3d_pw_dgrad (stationary)
]


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("shape", COEF_FOLD)
def test_pw_coef_fold(gpu, dtype, shape):
    """coef_fold: a consumer that derives its coefficient table from the BatchNorm-backward sums gives the SAME bits as the
    x3d_bn_bwd_finalize launch followed by the consumer reading the table -- dx, dW (through slabs: a fixed summation order),
    the per-(n, c) sums -- and its publishing launch leaves dgamma, dbeta and the table exactly as the finalize launch does."""
    ops = _ops()
    n, cin, cout, t, h, w, kind = shape
    g_ = _gen(71)
    gy, _ = rnd((n, cout, t, h, w), dtype, g_)
    yraw, _ = rnd((n, cout, t, h, w), dtype, g_)
    wt = torch.randn((cout, cin), generator=g_) * 0.2
    m_ = n * t * h * w
    sums = torch.stack([torch.randn(cout, generator=g_, dtype=torch.float64) * m_ ** 0.5,
                        torch.randn(cout, generator=g_, dtype=torch.float64) * m_ ** 0.5], 1).contiguous().to(gpu)
    mi = torch.stack([0.3 * torch.randn(cout, generator=g_), 0.5 + torch.rand(cout, generator=g_)], 1).contiguous().to(gpu)
    gamma = (1 + 0.3 * torch.randn(cout, generator=g_)).to(gpu)
    coef = torch.empty((cout, 4), device=gpu)
    dga_ref = torch.full((cout,), 0.25, device=gpu)
    dbe_ref = torch.full((cout,), 0.25, device=gpu)
    ops.bn_bwd_finalize(sums, float(m_), mi, gamma, coef, dga_ref, dbe_ref)
    dp = _panels(ops, wt, dtype, gpu)[1]
    dev = lambda v: None if v is None else v.to(gpu)
    gyd, yrd, wtd = dev(gy), dev(yraw), dev(wt)
    if kind in ("c", "pair_c"):
        braw, _ = rnd((n, cin, t, h, w), dtype, g_)
        bss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1)
        gate = torch.rand((n, cin), generator=g_)
        kw = dict(braw=dev(braw), b_ss=dev(bss), gate=dev(gate))
        x = None
    else:
        x, _ = rnd((n, cin, t, h, w), dtype, g_)
        x = dev(torch.relu(x))
        add, _ = rnd((n, cin, t, h, w), dtype, g_)
        kw = dict(add=dev(add))
    tc = dev(rnd((n, cin, t, h, w), dtype, g_)[0]) if kind == "a_tail" else None

    def run(fold):
        dx = torch.empty((n, cin, t, h, w), dtype=dtype, device=gpu)
        dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)
        out = dict(dx=dx, dw=dw)
        dga = torch.full((cout,), 0.25, device=gpu)
        dbe = torch.full((cout,), 0.25, device=gpu)
        cout_ = torch.zeros((cout, 4), device=gpu)
        pub = ops.bn_bwd_fold(sums, m_, mi, gamma, dga, dbe, cout_) if fold else None
        quiet = ops.bn_bwd_fold(sums, m_, mi, gamma) if fold else None
        cf = None if fold else coef
        if kind in ("c",):
            out["ncs"] = torch.zeros((n, cin, 2), dtype=torch.float64, device=gpu)
            assert ops.pw_bwd(gyd, yrd, cf, dp, dx, dw, ops.EPI_SWISH_BWD, nc_sums=out["ncs"], slab=True, coef_fold=pub, **kw)
        elif kind in ("a", "a_tail"):
            if tc is not None:
                out["ts"] = torch.zeros((cin, 2), dtype=torch.float64, device=gpu)
            assert ops.pw_bwd(gyd, yrd, cf, dp, dx, dw, ops.EPI_ADD, x=x, tail_c=tc, tail_sums_c=out.get("ts"), slab=True, coef_fold=pub, **kw)
        elif kind == "pair":
            assert ops.pw_wgrad(gyd, yrd, cf, x, dw, slab=True, coef_fold=quiet)
            ops.pw_dgrad(gyd, yrd, cf, wtd, dx, ops.EPI_ADD, w_panel=dp, coef_fold=pub, **kw)
        else:
            out["ncs"] = torch.zeros((n, cin, 2), dtype=torch.float64, device=gpu)
            assert ops.pw_wgrad(gyd, yrd, cf, kw["braw"], dw, in_ss=kw["b_ss"], in_gate=kw["gate"], in_act=2, slab=True, coef_fold=quiet)
            ops.pw_dgrad(gyd, yrd, cf, wtd, dx, ops.EPI_SWISH_BWD, nc_sums=out["ncs"], w_panel=dp, coef_fold=pub, **kw)
        torch.cuda.synchronize()
        out.update(dga=dga, dbe=dbe, coef=cout_)
        return out

    ref, got = run(False), run(True)
    assert torch.equal(got["dx"], ref["dx"]) and torch.equal(got["dw"], ref["dw"])
    # (the per-(n, c) sums and the tail sums are fp64 atomics over workgroups: summation order only)
    for k in ("ncs", "ts"):
        if k in ref:
            report(k, got[k], ref[k], 1e-12, 1e-9 * max(1.0, ref[k].abs().max().item()))
    assert torch.equal(got["coef"], coef), "the published table differs from x3d_bn_bwd_finalize's"
    assert torch.equal(got["dga"], dga_ref) and torch.equal(got["dbe"], dbe_ref)
    assert float((ref["dga"] - 0.25).abs().max()) == 0.0           # (without the fold the consumers leave dgamma alone)


@pytest.mark.parametrize("bf", HALF)
@pytest.mark.parametrize("shape", S.PW_BWD)   # N, Cin, Cout, T, H, W, epi   (a conv: Cin = block input, Cout = inner; c conv: Cin = inner, Cout = out)
def test_pw_bwd_fused(gpu, shape, bf):
    """x3d_pw_bwd (one pass over dY) against x3d_pw_dgrad + x3d_pw_wgrad: dx bit-identical (same bf16 operands,
    same accumulation order over Cout), dw and the per-(n,c) sums to fp32 summation-order tolerance."""
    ops = _ops()
    n, cin, cout, t, h, w, epi = shape
    g_ = _gen(21)
    wt = (torch.randn((cout, cin), generator=g_) * 0.2).to(gpu)
    (fp, dp), = ops.pw_pack_weights([wt], dtype=bf)
    gy = torch.randn((n, cout, t, h, w), generator=g_).to(bf).to(gpu)
    yraw = torch.randn((n, cout, t, h, w), generator=g_).to(bf).to(gpu)
    coef = (torch.randn((cout, 4), generator=g_) * 0.5).to(gpu)
    dx0 = torch.empty((n, cin, t, h, w), dtype=bf, device=gpu)
    dx1 = torch.empty_like(dx0)
    dw0 = torch.zeros((cout, cin), device=gpu)
    dw1 = torch.zeros_like(dw0)
    if epi == "swish_bwd":
        braw = torch.randn((n, cin, t, h, w), generator=g_).to(bf).to(gpu)
        bss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1).to(gpu)
        gate = torch.rand((n, cin), generator=g_).to(gpu)
        nc0 = torch.zeros((n, cin, 2), dtype=torch.float64, device=gpu)
        nc1 = torch.zeros_like(nc0)
        ops.pw_dgrad(gy, yraw, coef, wt, dx0, epi=ops.EPI_SWISH_BWD, braw=braw, b_ss=bss, gate=gate, nc_sums=nc0, w_panel=dp)
        ops.pw_wgrad(gy, yraw, coef, braw, dw0, in_ss=bss, in_gate=gate, in_act=2)
        ok = ops.pw_bwd(gy, yraw, coef, dp, dx1, dw1, ops.EPI_SWISH_BWD, braw=braw, b_ss=bss, gate=gate, nc_sums=nc1)
    else:
        x = torch.randn((n, cin, t, h, w), generator=g_).to(bf).to(gpu)
        if epi == "add":
            add = torch.randn((n, cin, t, h, w), generator=g_).to(bf).to(gpu)
            e = ops.EPI_ADD
        else:
            add = torch.randn((n, cin, t, (h + 1) // 2, (w + 1) // 2), generator=g_).to(bf).to(gpu)
            e = ops.EPI_ADD_STRIDED
        ops.pw_dgrad(gy, yraw, coef, wt, dx0, epi=e, add=add, w_panel=dp)
        ops.pw_wgrad(gy, yraw, coef, x, dw0)
        ok = ops.pw_bwd(gy, yraw, coef, dp, dx1, dw1, e, x=x, add=add)
    torch.cuda.synchronize()
    assert ok, "fused kernel should cover this shape"
    assert torch.equal(dx0, dx1)
    scale = dw0.abs().max().item()
    report("dw", dw1, dw0.double().cpu(), 2e-3, 2e-3 * scale)
    if epi == "swish_bwd":
        report("nc_sums", nc1, nc0.cpu(), 1e-4, 1e-4 * max(1.0, nc0.abs().max().item()))


@pytest.mark.parametrize("shape", [(2, 432, 192, 2, 8, 8), (3, 336, 72, 1, 8, 12), (2, 440, 200, 2, 4, 6),
                                   (3, 192, 432, 8, 7, 7), (2, 420, 180, 1, 8, 8), (40, 432, 192, 8, 7, 7),
                                   (2, 216, 96, 2, 14, 14), (2, 96, 216, 2, 14, 14), (24, 216, 96, 8, 14, 14), (2, 210, 90, 1, 8, 8),
                                   (3, 96, 432, 4, 14, 14), (20, 96, 432, 16, 14, 14),   # (forward: K = 96 -> M = 432)
                                   # X3D-XL widths on the weights-stationary kernel (row blocks sliced over blockIdx.y; K = 630 on four waves)
                                   (3, 630, 280, 2, 10, 10), (20, 630, 280, 16, 10, 10), (3, 280, 630, 2, 10, 10), (2, 306, 136, 4, 20, 20),
                                   (2, 136, 306, 4, 20, 20), (2, 162, 72, 2, 39, 39), (2, 72, 162, 4, 20, 20), (2, 136, 630, 2, 20, 20),
                                   (2, 72, 306, 2, 20, 20), (9, 306, 136, 16, 20, 20)])   # stage-5 block 0: dgrad K = 432 -> M = 96 with the strided add
@pytest.mark.parametrize("bf", HALF)
def test_pw_weights_streamed_path(gpu, shape, bf):
    """Deep, narrow layers (stage-5 shapes) with a packed panel run the weights-streamed 32-point-tile kernel
    (pw_gemm_ws.h) or, for K = 432 -> M <= 192 and K = 192 -> M <= 448, the weights-stationary one (pw_gemm_wst.h:
    (40, ...) and (24, ...) give every persistent workgroup several tiles across a sample boundary, the 7x7 ones a
    ragged last tile; K = 216 -> M <= 96 and K = 96 -> M <= 224 are the stage-4 forward instantiations); without a panel the same call runs the resident-panel kernel.  Same bf16 operands and the same
    accumulation order over K: outputs bit-identical; statistics / per-(n,c) sums to summation-order tolerance."""
    ops = _ops()
    n, cin, cout, t, h, w = shape
    g_ = _gen(41)
    wt = (torch.randn((cout, cin), generator=g_) * 0.1).to(gpu)
    (fp, dp), = ops.pw_pack_weights([wt], dtype=bf)
    x = torch.randn((n, cin, t, h, w), generator=g_).to(bf).to(gpu)
    ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1).to(gpu)
    gate = torch.rand((n, cin), generator=g_).to(gpu)
    for kw in (dict(), dict(in_ss=ss, in_gate=gate, in_act=2), dict(in_ss=ss, in_act=1)):
        s0 = torch.zeros((cout, 2), dtype=torch.float64, device=gpu)
        s1 = torch.zeros_like(s0)
        y0 = ops.pw_fwd(x, wt, stats=s0, **kw)
        y1 = ops.pw_fwd(x, wt, stats=s1, w_panel=fp, **kw)
        torch.cuda.synchronize()
        assert torch.equal(y0, y1)
        assert torch.allclose(s0, s1, rtol=1e-5, atol=1e-4)
    gy = torch.randn((n, cout, t, h, w), generator=g_).to(bf).to(gpu)
    yraw = torch.randn((n, cout, t, h, w), generator=g_).to(bf).to(gpu)
    coef = (torch.randn((cout, 4), generator=g_) * 0.5).to(gpu)
    add = torch.randn((n, cin, t, h, w), generator=g_).to(bf).to(gpu)
    braw = torch.randn((n, cin, t, h, w), generator=g_).to(bf).to(gpu)
    bss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1).to(gpu)
    add_half = torch.randn((n, cin, t, (h + 1) // 2, (w + 1) // 2), generator=g_).to(bf).to(gpu)   # strided shortcut gradient
    for kw in (dict(), dict(epi=ops.EPI_ADD, add=add), dict(epi=ops.EPI_SWISH_BWD, braw=braw, b_ss=bss, gate=gate),
               dict(epi=ops.EPI_ADD_STRIDED, add=add_half)):
        dx0, dx1 = torch.empty_like(x), torch.empty_like(x)
        extra0, extra1 = {}, {}
        if "braw" in kw:
            extra0["nc_sums"] = torch.zeros((n, cin, 2), dtype=torch.float64, device=gpu)
            extra1["nc_sums"] = torch.zeros((n, cin, 2), dtype=torch.float64, device=gpu)
        ops.pw_dgrad(gy, yraw, coef, wt, dx0, **kw, **extra0)
        ops.pw_dgrad(gy, yraw, coef, wt, dx1, w_panel=dp, **kw, **extra1)
        torch.cuda.synchronize()
        assert torch.equal(dx0, dx1)
        if extra0:
            assert torch.allclose(extra0["nc_sums"], extra1["nc_sums"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 24, 54, 4, 16, 16), (1, 96, 216, 2, 14, 14), (2, 200, 72, 2, 7, 7), (1, 432, 192, 3, 7, 7)])
@pytest.mark.parametrize("bf", HALF)
def test_pw_packed_panels(gpu, shape, bf):
    """bf16 GEMMs fed from x3d_pw_pack_weights panels give bit-identical results to the in-kernel fp32->bf16
    conversion (same rounded weights, same accumulation order), forward and dgrad, incl. row tiles past M."""
    ops = _ops()
    n, cin, cout, t, h, w = shape
    g_ = _gen(11)
    wt = (torch.randn((cout, cin), generator=g_) * 0.2).to(gpu)
    (fp, dp), = ops.pw_pack_weights([wt], dtype=bf)
    lib = __import__("x3d_tf_amd").hip.load()
    assert fp.numel() == lib.x3d_pw_panel_elems(cout, cin) and dp.numel() == lib.x3d_pw_panel_elems(cin, cout)
    pitch = (cin + 15) // 16 * 16 + 8
    rows, kp = (cout + 31) // 32 * 32, pitch - 8
    img = fp[:rows * pitch].view(rows, pitch).float().cpu()
    assert torch.equal(img[:cout, :cin], wt.to(bf).float().cpu())
    assert img[cout:].abs().sum().item() == 0 and img[:, cin:].abs().sum().item() == 0
    # second image: [row block][k-step][lane = 32 * half + r][8] = the 32x32x16 MFMA A operand, one 1 KB load per k-step
    tiled = fp[rows * pitch:].view(rows // 32, kp // 16, 2, 32, 8).float().cpu()
    want = img[:, :kp].view(rows // 32, 32, kp // 16, 2, 8).permute(0, 2, 3, 1, 4)
    assert torch.equal(tiled, want)
    x = torch.randn((n, cin, t, h, w), generator=g_).to(bf).to(gpu)
    ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1).to(gpu)
    y0 = ops.pw_fwd(x, wt, in_ss=ss, in_act=2)
    y1 = ops.pw_fwd(x, wt, in_ss=ss, in_act=2, w_panel=fp)
    gy = torch.randn((n, cout, t, h, w), generator=g_).to(bf).to(gpu)
    yraw = torch.randn((n, cout, t, h, w), generator=g_).to(bf).to(gpu)
    coef = (torch.randn((cout, 4), generator=g_) * 0.5).to(gpu)
    dx0, dx1 = torch.empty_like(x), torch.empty_like(x)
    ops.pw_dgrad(gy, yraw, coef, wt, dx0)
    ops.pw_dgrad(gy, yraw, coef, wt, dx1, w_panel=dp)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(dx0, dx1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", S.DW)   # N, C, T, H, W, stride
def test_dw3d_fwd(gpu, dtype, shape):
    ops, O = _ops(), _oracle()
    n, c, t, h, w, stride = shape
    g = _gen(2)
    x, xd = rnd((n, c, t, h, w), dtype, g)
    wt = torch.randn((c, 3, 3, 3), generator=g) * 0.3
    ss = torch.stack([1 + 0.3 * torch.randn(c, generator=g), 0.3 * torch.randn(c, generator=g)], 1)
    ref = O.depthwise3x3x3(_affine(xd, ss.double(), None, 1), wt.double(), stride)
    stats = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    pool = torch.zeros((n, c), dtype=torch.float64, device=gpu)
    y = ops.dw3d_fwd(x.to(gpu), wt.to(gpu), stride, in_ss=ss.to(gpu), in_act=1, stats=stats, pool=pool)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(ref.shape)
    rt, at = tol_store(dtype)   # inputs pre-rounded, fp32 arithmetic: only the output rounding differs
    mx = _dw_on_matrix_cores(S.dw_fwd_struct(shape, dtype))
    if mx:   # matrix-core kernel (dw_mx.hip): the fp32 prologue's output and the weights are rounded to the storage type first
        ref = O.depthwise3x3x3(round_to(_affine(xd, ss.double(), None, 1).float(), dtype), round_to(wt, dtype), stride)
        rt, at = tol_gemm(dtype)
    report("y", y, ref, rt, at * ref.abs().max().item())
    ys = y.float().cpu()
    sref = _stats_ref(ys, dtype)
    report("stats", stats, sref, _stol(dtype), _stol(dtype) * max(1.0, sref.abs().max().item()))
    report("pool", pool, ys.double().sum((2, 3, 4)), _stol(dtype), 10 * _stol(dtype) * max(1.0, float(ys[0, 0].numel()) ** 0.5))
    # no prologue
    y2 = ops.dw3d_fwd(x.to(gpu), wt.to(gpu), stride)
    ref2 = O.depthwise3x3x3(xd, round_to(wt, dtype) if mx else wt.double(), stride)
    # (the absolute part scales with THIS reference: the prologue form's can be tiny or all zeros -- one channel whose BN + ReLU clips
    # everything, fuzz seed 62 -- and sums of 27 products that cancel to 1e-5 of their terms need it)
    report("y_noprologue", y2, ref2, rt, at * ref2.abs().max().item())


@pytest.mark.parametrize("panel", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", S.PW_DGRAD)   # N, Cin, Cout, T, H, W
@pytest.mark.parametrize("epi", S.PW_DGRAD_EPI)
def test_pw_dgrad(gpu, dtype, shape, epi, panel):
    if panel and dtype == torch.float32:
        pytest.skip("weight panels exist for the 16-bit storage types only")
    ops = _ops()
    n, cin, cout, t, h, w = shape
    g_ = _gen(3)
    g, gd = rnd((n, cout, t, h, w), dtype, g_)
    yraw, yd = rnd((n, cout, t, h, w), dtype, g_)
    coef = torch.randn((cout, 4), generator=g_) * 0.5
    wt = torch.randn((cout, cin), generator=g_) * 0.2
    dyraw = _dyraw(coef, gd, yd, dtype)       # operands as the matrix cores see them (fp32 prologue, rounded to storage)
    ref = torch.einsum("oc,nothw->ncthw", round_to(wt, dtype), dyraw)
    dx = torch.empty((n, cin, t, h, w), dtype=dtype, device=gpu)
    kw = {}
    rt, at = tol_gemm(dtype)
    if epi == "add":
        add, addd = rnd((n, cin, t, h, w), dtype, g_)
        ref = ref + addd
        kw = dict(epi=ops.EPI_ADD, add=add.to(gpu))
    elif epi == "add_strided":
        hh, wh = (h + 1) // 2, (w + 1) // 2
        add, addd = rnd((n, cin, t, hh, wh), dtype, g_)
        up = torch.zeros_like(ref)
        up[:, :, :, ::2, ::2] = addd
        ref = ref + up
        kw = dict(epi=ops.EPI_ADD_STRIDED, add=add.to(gpu))
    elif epi == "swish_bwd":
        braw, bd = rnd((n, cin, t, h, w), dtype, g_)
        bss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1)
        gate = torch.rand((n, cin), generator=g_)
        v = _affine(bd, bss.double(), gate.double(), 0)
        s = torch.sigmoid(v)
        ref = ref * (s * (1 + v * (1 - s)))
        ncs = torch.zeros((n, cin, 2), dtype=torch.float64, device=gpu)
        kw = dict(epi=ops.EPI_SWISH_BWD, braw=braw.to(gpu), b_ss=bss.to(gpu), gate=gate.to(gpu), nc_sums=ncs)
    if panel:
        kw["w_panel"] = _panels(ops, wt, dtype, gpu)[1]     # the production dispatch (model.py always passes the panel)
    ops.pw_dgrad(g.to(gpu), yraw.to(gpu), coef.to(gpu), wt.to(gpu), dx, **kw)
    torch.cuda.synchronize()
    report("dx", dx, ref, rt, at * ref.abs().max().item() + 1.1 * _dyraw_slack(coef, gd, yd, dtype, wt))
    if epi == "swish_bwd":
        dvs = dx.float().cpu().double()
        sref = torch.stack([dvs.sum((2, 3, 4)), (dvs * bd).sum((2, 3, 4))], -1)
        report("nc_sums", ncs, sref, 10 * _stol(dtype), 10 * _stol(dtype) * max(1.0, sref.abs().max().item()))


@pytest.mark.parametrize("slab", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", S.PW_WGRAD)   # N, Cin, Cout, T, H, W, stride, prologue
def test_pw_wgrad(gpu, dtype, shape, slab):
    """slab: through per-point-chunk partial slabs + x3d_dw_slab_reduce (x3d_hip.h dw_slab), where the kernel behind the call
    has the form (the 12-tile groups: the stage-5 layers)."""
    ops = _ops()
    n, cin, cout, t, h, w, stride, pro = shape
    if slab and not hip_lib().x3d_pw_wgrad_dw_parts(S.pw_wgrad_struct(shape, dtype)):
        pytest.skip("no slab form behind this call")
    g_ = _gen(4)
    ho, wo = -(-h // stride), -(-w // stride)
    x, xd = rnd((n, cin, t, h, w), dtype, g_)
    g, gd = rnd((n, cout, t, ho, wo), dtype, g_)
    yraw, yd = rnd((n, cout, t, ho, wo), dtype, g_)
    coef = torch.randn((cout, 4), generator=g_) * 0.5
    dyraw = _dyraw(coef, gd, yd, dtype)       # both operands as the matrix cores see them
    ss = gate = None
    act = 0
    xin = xd
    if pro:
        ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1)
        gate = torch.rand((n, cin), generator=g_)
        act = 2
        xin = round_to(_affine(x.float(), ss, gate, act), dtype)
    if stride > 1:
        xin = xin[:, :, :, ::stride, ::stride]
    ref = torch.einsum("nothw,ncthw->oc", dyraw, xin)
    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)   # += semantics
    assert ops.pw_wgrad(g.to(gpu), yraw.to(gpu), coef.to(gpu), x.to(gpu), dw, in_ss=None if ss is None else ss.to(gpu),
                        in_gate=None if gate is None else gate.to(gpu), in_act=act, stride=stride, slab=slab)
    torch.cuda.synchronize()
    tol = _wtol(dtype)
    report("dw", dw, ref + 0.5, tol, tol * ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", S.DW)
def test_dw3d_bwd(gpu, dtype, shape):
    ops, O = _ops(), _oracle()
    n, c, t, h, w, stride = shape
    g_ = _gen(5)
    ho, wo = -(-h // stride), -(-w // stride)
    araw, ad = rnd((n, c, t, h, w), dtype, g_)
    dv, dvd = rnd((n, c, t, ho, wo), dtype, g_)
    braw, bd = rnd((n, c, t, ho, wo), dtype, g_)
    coef = torch.randn((n, c, 4), generator=g_) * 0.5
    wt = torch.randn((c, 3, 3, 3), generator=g_) * 0.3
    ss = torch.stack([1 + 0.3 * torch.randn(c, generator=g_), 0.3 * torch.randn(c, generator=g_)], 1)
    cd = coef.double()
    dB = cd[:, :, 0, None, None, None] * dvd + cd[:, :, 1, None, None, None] * bd + cd[:, :, 2, None, None, None]
    z = _affine(ad, ss.double())
    mx = _dw_on_matrix_cores(S.dw_bwd_struct(shape, dtype))
    if mx:   # matrix-core kernel (dw_mx.hip): dB, A = relu(z) and the weights enter the products rounded to the storage type
        dB = round_to(dB.float(), dtype)
        act = round_to(F.relu(z).float(), dtype).requires_grad_(True)
        wref = round_to(wt, dtype).requires_grad_(True)
    else:
        act = F.relu(z).requires_grad_(True)
        wref = wt.double().requires_grad_(True)
    out = O.depthwise3x3x3(act, wref, stride)
    dA, dWr = torch.autograd.grad((out * dB).sum(), [act, wref])
    ga_ref = dA * (z > 0)
    ga = torch.empty((n, c, t, h, w), dtype=dtype, device=gpu)
    a_sums = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    dw = torch.full((c, 27), 0.25, dtype=torch.float32, device=gpu)
    ops.dw3d_bwd(dv.to(gpu), braw.to(gpu), coef.to(gpu), araw.to(gpu), ss.to(gpu), wt.to(gpu).view(c, 27), ga,
                 a_sums, dw, stride)
    torch.cuda.synchronize()
    rt, at = tol_gemm(dtype) if mx else tol_store(dtype)
    report("ga", ga, ga_ref, rt, at * ga_ref.abs().max().item())
    wtol = _wtol(dtype) if mx else 2e-4
    report("dw", dw, dWr.view(c, 27) + 0.25, wtol, wtol * dWr.abs().max().item())
    gs = ga.float().cpu().double()
    sref = torch.stack([gs.sum((0, 2, 3, 4)), (gs * ad).sum((0, 2, 3, 4))], 1)
    report("a_sums", a_sums, sref, 10 * _stol(dtype), 10 * _stol(dtype) * max(1.0, sref.abs().max().item()))


# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 4, 16, 16, 24), (1, 3, 9, 11, 24), (1, 2, 20, 20, 32),
                                   (1, 2, 18, 160, 24), (2, 3, 13, 224, 24),    # wgrad fast path: 2 segments / row, odd H
                                   (1, 2, 7, 312, 24), (2, 2, 6, 24, 24)])      # W % 16 == 8 (X3D-L / XL clips): half a vector at the row end
def test_stem(gpu, dtype, shape):
    ops = _ops()
    n, t, h, w, c1 = shape
    g_ = _gen(6)
    x, xd = rnd((n, 3, t, h, w), dtype, g_)
    ws = torch.randn((c1, 3, 3, 3), generator=g_) * 0.3
    wt = torch.randn((c1, 5), generator=g_) * 0.4
    ref_s = F.conv3d(F.pad(xd, (1, 1, 1, 1, 0, 0)), ws.double().unsqueeze(2), stride=(1, 2, 2))
    ys = ops.stem_s_fwd(x.to(gpu), ws.to(gpu))
    torch.cuda.synchronize()
    # 16-bit storage: the matrix-core path (W % 8 == 0) rounds the weights to the storage type, the scalar path keeps them
    # in fp32 -- the output must match the fp64 conv over ONE of the two weight sets to the output rounding
    ref_r = F.conv3d(F.pad(xd, (1, 1, 1, 1, 0, 0)), round_to(ws, dtype).unsqueeze(2), stride=(1, 2, 2))
    rt, at = tol_gemm(dtype)
    try:
        report("conv_s", ys, ref_r, rt, at * ref_s.abs().max().item())
    except AssertionError:
        report("conv_s", ys, ref_s, rt, at * ref_s.abs().max().item())
    # conv_t on the stored conv_s output
    ysd = ys.float().cpu().double()
    ref_t = F.conv3d(F.pad(ysd, (0, 0, 0, 0, 2, 2)), wt.double().view(-1, 1, 5, 1, 1), groups=c1)
    stats = torch.zeros((c1, 2), dtype=torch.float64, device=gpu)
    yt = ops.dwt_fwd(ys, wt.to(gpu), stats=stats)
    torch.cuda.synchronize()
    rs, as_ = tol_store(dtype)
    report("conv_t", yt, ref_t, rs, as_ * ref_t.abs().max().item())
    report("stats", stats, _stats_ref(yt.float().cpu(), dtype), 1e-5, 1e-4)
    # inference epilogue: the stem's BatchNorm (moving statistics) + ReLU on the accumulator, no raw tensor stored
    oss = torch.stack([1 + 0.3 * torch.randn(c1, generator=g_), 0.3 * torch.randn(c1, generator=g_)], 1)
    yi = ops.dwt_fwd(ys, wt.to(gpu), out_ss=oss.to(gpu), out_act=1)
    ref_i = F.relu(_affine(ref_t, oss.double()))
    report("conv_t + bn + relu", yi, ref_i, rs, as_ * max(1.0, ref_i.abs().max().item()))
    yi0 = ops.dwt_fwd(ys, wt.to(gpu), out_ss=oss.to(gpu), out_act=0)
    report("conv_t + bn", yi0, _affine(ref_t, oss.double()), rs, as_ * max(1.0, ref_i.abs().max().item()))
    # backward of conv_t (+BN coefficients) and wgrad of conv_s
    g, gd = rnd(tuple(yt.shape), dtype, g_)
    coef = torch.randn((c1, 4), generator=g_) * 0.5
    ytd = yt.float().cpu().double()
    dY = (coef[:, 0].double().view(1, -1, 1, 1, 1) * gd + coef[:, 1].double().view(1, -1, 1, 1, 1) * ytd
          + coef[:, 2].double().view(1, -1, 1, 1, 1))
    xs = ysd.clone().requires_grad_(True)
    wtr = wt.double().requires_grad_(True)
    out = F.conv3d(F.pad(xs, (0, 0, 0, 0, 2, 2)), wtr.view(-1, 1, 5, 1, 1), groups=c1)
    dxs, dwt_ref = torch.autograd.grad((out * dY).sum(), [xs, wtr])
    dx = torch.empty_like(ys)
    dwt = torch.zeros((c1, 5), dtype=torch.float32, device=gpu)
    ops.dwt_bwd(g.to(gpu), yt, coef.to(gpu), ys, wt.to(gpu), dx, dwt)
    torch.cuda.synchronize()
    report("dwt_dx", dx, dxs, rs, as_ * dxs.abs().max().item())
    report("dwt_dw", dwt, dwt_ref, 2e-4, 2e-4 * dwt_ref.abs().max().item())
    # the same launch with the ReLU mask applied inside (relu_scale_shift): bit-identical to masking first, and
    # x3d_relu_bn_bwd_reduce with g = NULL returns the same sums as the storing form
    rss = torch.stack([1 + 0.3 * torch.randn(c1, generator=g_), 0.3 * torch.randn(c1, generator=g_)], 1).to(gpu)
    gg = g.to(gpu)
    gm = torch.empty_like(gg)
    s_store = torch.zeros((c1, 2), dtype=torch.float64, device=gpu)
    s_only = torch.zeros((c1, 2), dtype=torch.float64, device=gpu)
    ops.relu_bn_bwd_reduce(gg, None, yt, rss, gm, s_store)
    ops.relu_bn_bwd_reduce(gg, None, yt, rss, None, s_only)
    dx_a, dx_b = torch.empty_like(ys), torch.empty_like(ys)
    dw_a = torch.zeros((c1, 5), dtype=torch.float32, device=gpu)
    dw_b = torch.zeros((c1, 5), dtype=torch.float32, device=gpu)
    ops.dwt_bwd(gm, yt, coef.to(gpu), ys, wt.to(gpu), dx_a, dw_a)
    ops.dwt_bwd(gg, yt, coef.to(gpu), ys, wt.to(gpu), dx_b, dw_b, relu_ss=rss)
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_b), "mask inside x3d_dwt_bwd differs from masking first"
    report("dwt_dw masked", dw_b, dw_a, 1e-5, 1e-5 * dw_a.abs().max().item())
    report("reduce-only sums", s_only, s_store, 1e-9, 1e-9 * max(1.0, s_store.abs().max().item()))
    zmask = (rss[:, 0].view(1, -1, 1, 1, 1) * yt.float() + rss[:, 1].view(1, -1, 1, 1, 1)) > 0
    assert torch.equal(gm, torch.where(zmask, gg, torch.zeros_like(gg)))
    dxsd = dx.float().cpu().double()
    wsr = ws.double().requires_grad_(True)
    out_s = F.conv3d(F.pad(xd, (1, 1, 1, 1, 0, 0)), wsr.unsqueeze(2), stride=(1, 2, 2))
    (dws_ref,) = torch.autograd.grad((out_s * dxsd).sum(), [wsr])
    dws = torch.zeros((c1, 3, 3, 3), dtype=torch.float32, device=gpu)
    ops.stem_s_wgrad(x.to(gpu), dx, dws)
    torch.cuda.synchronize()
    report("stem_s_dw", dws, dws_ref, 2e-4, 2e-4 * dws_ref.abs().max().item())
    # the clip batch channels-last, as the reference's model takes it (X3D_LAYOUT_NTHWC): the matrix-core kernels read it in
    # place -- same products in the same order as the planar form, so the outputs are bit-identical
    from x3d_tf_amd import hip
    if hip.load().x3d_stem_s_nthwc_supported(3, w, c1, hip.dtype_code(dtype)):
        xcl = x.to(gpu).permute(0, 2, 3, 4, 1).contiguous()
        ys_cl = ops.stem_s_fwd(xcl, ws.to(gpu), channels_last=True)
        dws_cl = torch.zeros((c1, 3, 3, 3), dtype=torch.float32, device=gpu)
        ops.stem_s_wgrad(xcl, dx, dws_cl, channels_last=True)
        torch.cuda.synchronize()
        assert torch.equal(ys_cl, ys), "channels-last stem forward differs from the planar form"
        report("stem_s_dw channels-last", dws_cl, dws_ref, 2e-4, 2e-4 * dws_ref.abs().max().item())
    else:
        with pytest.raises(Exception):
            ops.stem_s_fwd(x.to(gpu).permute(0, 2, 3, 4, 1).contiguous(), ws.to(gpu), channels_last=True)


STEM_FUSED = [(2, 4, 16, 16, 24), (1, 16, 12, 224, 24), (1, 2, 10, 312, 24),     # two segments per row (64 + 48 columns), T = 16: three ring turns
              (3, 5, 10, 72, 24),                           # 15 segments: the last group is three segments, Wo = 36 of 64 columns
              (1, 7, 6, 312, 32), (2, 3, 5, 160, 32),       # X3D-XL: 32 channels (four rows per wave), Wo % 8 == 4, odd H
              (1, 1, 8, 24, 24), (2, 2, 9, 32, 8),          # T < KT; Cout = 8 (rows 8.. of every wave idle)
              (5, 6, 40, 128, 24)]                          # 100 segments = 25 groups: several groups per workgroup on a small GPU grid


@pytest.mark.parametrize("dtype", S.HALF_DTYPES)
@pytest.mark.parametrize("shape", STEM_FUSED)
def test_stem_fused(gpu, dtype, shape):
    """x3d_stem_fwd / x3d_stem_bwd (conv_s -> conv_t in one launch each way, reference model.py:202-206) against the
    two-kernel path they replace -- the forward BIT FOR BIT (same products in the same order, the conv_s output rounded to
    the storage type on chip as it was in HBM) -- and the two weight gradients against fp64 autograd as well."""
    ops = _ops()
    n, t, h, w, c1 = shape
    g_ = _gen(11)
    x, xd = rnd((n, 3, t, h, w), dtype, g_)
    ws = (torch.randn((c1, 3, 3, 3), generator=g_) * 0.3).to(gpu)
    wt = (torch.randn((c1, 5), generator=g_) * 0.4).to(gpu)
    xcl = x.to(gpu).permute(0, 2, 3, 4, 1).contiguous()
    assert ops.stem_fused_supported(xcl, c1) == (3 if c1 <= 24 else 1)     # (the backward kernel runs with 32 channels too: checked below)
    ys = ops.stem_s_fwd(xcl, ws, channels_last=True)
    st_ref = torch.zeros((c1, 2), dtype=torch.float64, device=gpu)
    yt_ref = ops.dwt_fwd(ys, wt, stats=st_ref)
    st = torch.zeros((c1, 2), dtype=torch.float64, device=gpu)
    yt = ops.stem_fwd(xcl, ws, wt, stats=st)
    torch.cuda.synchronize()
    assert torch.equal(yt, yt_ref), f"fused stem forward differs from conv_s + conv_t: {(yt.float() - yt_ref.float()).abs().max().item()}"
    report("stats", st, st_ref, 1e-5, 1e-5 * max(1.0, st_ref.abs().max().item()))
    # inference epilogue
    oss = torch.stack([1 + 0.3 * torch.randn(c1, generator=g_), 0.3 * torch.randn(c1, generator=g_)], 1).to(gpu)
    for act in (0, 1):
        yi = ops.stem_fwd(xcl, ws, wt, out_ss=oss, out_act=act)
        yi_ref = ops.dwt_fwd(ys, wt, out_ss=oss, out_act=act)
        torch.cuda.synchronize()
        assert torch.equal(yi, yi_ref), f"fused stem inference epilogue (act {act}) differs"
    # backward: unmasked gradient + mask inside, as the train plan runs it
    g, gd = rnd(tuple(yt.shape), dtype, g_)
    coef = (torch.randn((c1, 4), generator=g_) * 0.5).to(gpu)
    rss = torch.stack([1 + 0.3 * torch.randn(c1, generator=g_), 0.3 * torch.randn(c1, generator=g_)], 1).to(gpu)
    gg = g.to(gpu)
    for relu_ss in (None, rss):
        ds = torch.empty_like(ys)
        dwt_ref = torch.zeros((c1, 5), dtype=torch.float32, device=gpu)
        dws_ref = torch.zeros((c1, 3, 3, 3), dtype=torch.float32, device=gpu)
        ops.dwt_bwd(gg, yt, coef, ys, wt, ds, dwt_ref, relu_ss=relu_ss)
        ops.stem_s_wgrad(xcl, ds, dws_ref, channels_last=True)
        dwt = torch.full((c1, 5), 0.5, dtype=torch.float32, device=gpu)          # += on what is there
        dws = torch.full((c1, 3, 3, 3), -0.25, dtype=torch.float32, device=gpu)
        ops.stem_bwd(gg, yt, coef, xcl, ws, wt, dws, dwt, relu_ss=relu_ss)
        torch.cuda.synchronize()
        report("dw_t vs two kernels", dwt - 0.5, dwt_ref, 2e-5, 2e-5 * dwt_ref.abs().max().item())
        report("dw_s vs two kernels", dws + 0.25, dws_ref, 2e-5, 2e-5 * dws_ref.abs().max().item())
    # fp64 autograd over the stored tensors (mask inside): dY -> conv_t^T -> ds rounded to storage -> conv_s weight gradient
    ysd, ytd = ys.float().cpu().double(), yt.float().cpu().double()
    cf, rs_ = coef.cpu().double(), rss.cpu().double()
    v = lambda a: a.view(1, -1, 1, 1, 1)
    gm = torch.where(v(rs_[:, 0]) * ytd + v(rs_[:, 1]) > 0, gd, torch.zeros_like(gd))
    dY = v(cf[:, 0]) * gm + v(cf[:, 1]) * ytd + v(cf[:, 2])
    xs = ysd.clone().requires_grad_(True)
    wtr = wt.cpu().double().requires_grad_(True)
    out = F.conv3d(F.pad(xs, (0, 0, 0, 0, 2, 2)), wtr.view(-1, 1, 5, 1, 1), groups=c1)
    dxs, dwt64 = torch.autograd.grad((out * dY).sum(), [xs, wtr])
    report("dw_t vs fp64", dwt - 0.5, dwt64, 2e-4, 2e-4 * dwt64.abs().max().item())
    wsr = ws.cpu().double().requires_grad_(True)
    out_s = F.conv3d(F.pad(xd, (1, 1, 1, 1, 0, 0)), wsr.unsqueeze(2), stride=(1, 2, 2))
    (dws64,) = torch.autograd.grad((out_s * round_to(dxs.float(), dtype)).sum(), [wsr])
    report("dw_s vs fp64", dws + 0.25, dws64, 2e-4, 2e-4 * dws64.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 24, 3, 112, 112), (1, 24, 2, 56, 56), (2, 48, 3, 28, 28), (1, 96, 5, 14, 14),     # X3D-M shortcuts
                                   (1, 32, 2, 156, 156), (1, 32, 2, 78, 78), (2, 72, 1, 39, 39), (1, 136, 3, 20, 20),   # X3D-XL / L: odd 39
                                   (3, 5, 2, 7, 9), (1, 3, 1, 1, 1), (2, 2, 3, 40, 24)])
def test_subsample2(gpu, dtype, shape):
    """x3d_subsample2: the even-pixel copy a strided shortcut conv reads (reference model.py:360-367 samples pixels 0, 2, 4, ...):
    a pure copy, so bit-exact, on every vector width the launcher picks (16 / 8 / 4-byte loads and element loads, odd extents)."""
    ops = _ops()
    g_ = _gen(41)
    x, _ = rnd(shape, dtype, g_)
    xg = x.to(gpu)
    out = ops.subsample2(xg)
    torch.cuda.synchronize()
    assert torch.equal(out, xg[..., ::2, ::2].contiguous())
    buf = torch.empty(xg.numel() + 8, dtype=dtype, device=gpu)      # a 2- / 4-byte aligned source: the narrower vector paths
    sl = buf[1:1 + xg.numel()].view(shape)
    sl.copy_(xg)
    assert sl.data_ptr() % 16 != 0 and torch.equal(ops.subsample2(sl), sl[..., ::2, ::2].contiguous())


def test_stem_fused_unsupported(gpu):
    """fp32 storage, a planar batch, W % 8 != 0 and KT != 5 stay on the two-kernel path: the fused entry points refuse them."""
    from x3d_tf_amd import hip
    lib = hip.load()
    assert lib.x3d_stem_fused_supported(3, 24, 5, 2, 4, 16, 16, hip.dtype_code(torch.bfloat16), 1) == 3
    assert lib.x3d_stem_fused_supported(3, 32, 5, 2, 4, 16, 16, hip.dtype_code(torch.float16), 1) == 1      # forward only
    assert lib.x3d_stem_fused_supported(3, 24, 5, 2, 4, 16, 16, hip.dtype_code(torch.float32), 1) == 0
    assert lib.x3d_stem_fused_supported(3, 24, 5, 2, 4, 16, 16, hip.dtype_code(torch.bfloat16), 0) == 0
    assert lib.x3d_stem_fused_supported(3, 24, 5, 2, 4, 16, 20, hip.dtype_code(torch.bfloat16), 1) == 0
    assert lib.x3d_stem_fused_supported(3, 24, 3, 2, 4, 16, 16, hip.dtype_code(torch.bfloat16), 1) == 0
    assert lib.x3d_stem_fused_supported(3, 40, 5, 2, 4, 16, 16, hip.dtype_code(torch.bfloat16), 1) == 0
    ops = _ops()
    x = torch.randn(1, 2, 8, 16, 3, device=gpu)
    with pytest.raises(hip.X3DHipError):
        ops.stem_fwd(x, torch.randn(24, 3, 3, 3, device=gpu), torch.randn(24, 5, device=gpu))


@pytest.mark.parametrize("dtype", S.HALF_DTYPES)
@pytest.mark.parametrize("n,c,P,conv", [(23, 5, (16, 7, 7), True), (23, 3, (16, 7, 7), False), (64, 2, (16, 7, 7), True),
                                        (7, 4, (4, 5, 10), True), (33, 2, (2, 8, 8), False)])
def test_tail_bwd_grouped(gpu, dtype, n, c, P, conv):
    """x3d_tail_bwd on small planes in 16-bit storage (tail_bwd_small_kernel, elem.hip): NB samples of one channel per
    workgroup, grid = (C, ceil(N / NB)) -- here with N > NB and a PARTIAL last group (the X3D-M stage-5 case is N = 64,
    P = 16x7x7: NB = 10 -> six full groups and one of four; 23 -> 10 + 10 + 3).  The masked gradient is exact (a select), so
    a sample-indexing error in dy * [y > 0] of any group shows as a bit difference; the sums are fp64 reductions."""
    ops = _ops()
    g_ = _gen(31)
    shape = (n, c) + P
    craw, cd = rnd(shape, dtype, g_)
    rraw, rd = rnd(shape, dtype, g_)
    y, yd = rnd(shape, dtype, g_)        # the block output: only its sign pattern matters here
    dy, dyd = rnd(shape, dtype, g_)
    dyg = dy.to(gpu).clone()
    sc = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    sr = torch.zeros((c, 2), dtype=torch.float64, device=gpu) if conv else None
    ops.tail_bwd(dyg, y.to(gpu), craw.to(gpu), rraw.to(gpu) if conv else None, sc, sr)
    torch.cuda.synchronize()
    gref = dyd * (yd > 0)
    report("tail_bwd_g", dyg, gref, 0, 0)
    for nn in range(n):                  # per sample, so that a failure names the group it sits in
        assert torch.equal(dyg[nn].float().cpu().double(), gref[nn]), f"sample {nn} of {n}"
    report("sums_c", sc, torch.stack([gref.sum((0, 2, 3, 4)), (gref * cd).sum((0, 2, 3, 4))], 1), 1e-5, 1e-4)
    if conv:
        report("sums_r", sr, torch.stack([gref.sum((0, 2, 3, 4)), (gref * rd).sum((0, 2, 3, 4))], 1), 1e-5, 1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("P", [(2, 4, 8), (13, 5, 5)])
def test_bn_tail_pool(gpu, dtype, P):
    ops = _ops()
    n, c = 3, 10
    g_ = _gen(7)
    shape = (n, c) + P
    craw, cd = rnd(shape, dtype, g_)
    rraw, rd = rnd(shape, dtype, g_)
    M = n * P[0] * P[1] * P[2]
    # finalize from exact statistics
    stats = torch.stack([cd.sum((0, 2, 3, 4)), (cd * cd).sum((0, 2, 3, 4))], 1).to(gpu)
    gamma = (1 + 0.2 * torch.randn(c, generator=g_))
    beta = 0.2 * torch.randn(c, generator=g_)
    mm, mv = torch.zeros(c), torch.ones(c)
    ss = torch.empty((c, 2), device=gpu)
    mi = torch.empty((c, 2), device=gpu)
    mmg, mvg = mm.to(gpu), mv.to(gpu)
    ops.bn_finalize(stats, M, gamma.to(gpu), beta.to(gpu), mmg, mvg, 1e-5, 0.9, True, ss, mi)
    mean = cd.mean((0, 2, 3, 4))
    var = cd.var((0, 2, 3, 4), unbiased=False)
    inv = 1 / torch.sqrt(var + 1e-5)
    report("scale", ss[:, 0], gamma.double() * inv, 1e-5, 1e-6)
    report("shift", ss[:, 1], beta.double() - mean * gamma.double() * inv, 1e-5, 1e-6)
    report("moving_mean", mmg, 0.1 * mean, 1e-5, 1e-7)
    report("moving_var", mvg, 0.9 + 0.1 * var * M / (M - 1), 1e-5, 1e-7)
    ss2 = torch.empty((c, 2), device=gpu)
    mi2 = torch.empty((c, 2), device=gpu)
    ops.bn_eval_coef(gamma.to(gpu), beta.to(gpu), mmg, mvg, 1e-5, ss2, mi2)
    inv2 = 1 / torch.sqrt(mvg.cpu().double() + 1e-5)
    report("eval_scale", ss2[:, 0], gamma.double() * inv2, 1e-5, 1e-6)
    # tail fwd (conv shortcut and identity)
    ssr = torch.stack([1 + 0.3 * torch.randn(c, generator=g_), 0.3 * torch.randn(c, generator=g_)], 1)
    ssc = ss.cpu().double()
    rt, at = tol_store(dtype)
    y = torch.empty(shape, dtype=dtype, device=gpu)
    ops.tail_fwd(craw.to(gpu), ss, rraw.to(gpu), ssr.to(gpu), y)
    ref = F.relu(_affine(cd, ssc) + _affine(rd, ssr.double()))
    report("tail_conv", y, ref, rt, at * ref.abs().max().item())
    y2 = torch.empty(shape, dtype=dtype, device=gpu)
    ops.tail_fwd(craw.to(gpu), ss, rraw.to(gpu), None, y2)
    ref2 = F.relu(_affine(cd, ssc) + rd)
    report("tail_identity", y2, ref2, rt, at * ref2.abs().max().item())
    # tail bwd
    dy, dyd = rnd(shape, dtype, g_)
    dyg = dy.to(gpu).clone()
    sc = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    sr = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    ops.tail_bwd(dyg, y, craw.to(gpu), rraw.to(gpu), sc, sr)
    yd = y.float().cpu().double()
    gref = dyd * (yd > 0)
    report("tail_bwd_g", dyg, gref, 0, 0)
    report("sums_c", sc, torch.stack([gref.sum((0, 2, 3, 4)), (gref * cd).sum((0, 2, 3, 4))], 1), 1e-5, 1e-4)
    report("sums_r", sr, torch.stack([gref.sum((0, 2, 3, 4)), (gref * rd).sum((0, 2, 3, 4))], 1), 1e-5, 1e-4)
    # bn backward finalize: check the coefficients reproduce the BN gradient
    coef = torch.empty((c, 4), device=gpu)
    dgam = torch.zeros(c, device=gpu)
    dbet = torch.zeros(c, device=gpu)
    ops.bn_bwd_finalize(sc, M, mi, gamma.to(gpu), coef, dgam, dbet)
    x_ = cd.clone().requires_grad_(True)
    gm = gamma.double().requires_grad_(True)
    bt = beta.double().requires_grad_(True)
    mu = x_.mean((0, 2, 3, 4), keepdim=True)
    vr = x_.var((0, 2, 3, 4), unbiased=False, keepdim=True)
    bn = (x_ - mu) / torch.sqrt(vr + 1e-5) * gm.view(1, -1, 1, 1, 1) + bt.view(1, -1, 1, 1, 1)
    dx_ref, dg_ref, db_ref = torch.autograd.grad((bn * gref).sum(), [x_, gm, bt])
    cf = coef.cpu().double()
    dx_got = cf[:, 0].view(1, -1, 1, 1, 1) * gref + cf[:, 1].view(1, -1, 1, 1, 1) * cd + cf[:, 2].view(1, -1, 1, 1, 1)
    report("bn_bwd_dx", dx_got, dx_ref, 1e-4, 1e-5)
    report("bn_bwd_dgamma", dgam, dg_ref, 1e-4, 1e-4)
    report("bn_bwd_dbeta", dbet, db_ref, 1e-4, 1e-4)
    # relu+bn backward reduce (both sources) and pool
    gbuf = torch.empty(shape, dtype=dtype, device=gpu)
    s2 = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    ops.relu_bn_bwd_reduce(dy.to(gpu), None, craw.to(gpu), ss, gbuf, s2)
    z = _affine(cd, ssc)
    gr = dyd * (z > 0)
    report("rbr_g", gbuf, gr, 0, 0)
    report("rbr_sums", s2, torch.stack([gr.sum((0, 2, 3, 4)), (gr * cd).sum((0, 2, 3, 4))], 1), 1e-5, 1e-4)
    dpool = torch.randn((n, c), generator=g_)
    s3 = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    ops.relu_bn_bwd_reduce(None, dpool.to(gpu), craw.to(gpu), ss, gbuf, s3)
    pp = P[0] * P[1] * P[2]
    gr3 = (dpool.double() / pp)[:, :, None, None, None] * (z > 0)
    report("rbr_pool_g", gbuf, gr3, rt, at * gr3.abs().max().item())
    pooled = torch.empty((n, c), device=gpu)
    ops.pool_fwd(craw.to(gpu), ss, pooled)
    report("pool", pooled, F.relu(z).mean((2, 3, 4)), 1e-5, 1e-5)


def test_se(gpu):
    ops = _ops()
    n, c, wd, P = 3, 54, 8, 640.0
    g_ = _gen(8)
    pool_sums = torch.randn((n, c), generator=g_, dtype=torch.float64) * 30
    bss = torch.stack([1 + 0.3 * torch.randn(c, generator=g_), 0.3 * torch.randn(c, generator=g_)], 1)
    w1 = torch.randn((wd, c), generator=g_) * 0.3
    b1 = torch.randn(wd, generator=g_) * 0.1
    w2 = torch.randn((c, wd), generator=g_) * 0.3
    b2 = torch.randn(c, generator=g_) * 0.1
    gate = torch.empty((n, c), device=gpu)
    hidden = torch.empty((n, wd), device=gpu)
    ops.se_fwd(pool_sums.to(gpu), P, bss.to(gpu), w1.to(gpu), b1.to(gpu), w2.to(gpu), b2.to(gpu), gate, hidden)
    pooled = bss[:, 0].double() * pool_sums / P + bss[:, 1].double()
    h = F.relu(pooled @ w1.double().t() + b1.double())
    gr = torch.sigmoid(h @ w2.double().t() + b2.double())
    report("hidden", hidden, h, 1e-5, 1e-5)
    report("gate", gate, gr, 1e-5, 1e-5)


@pytest.mark.parametrize("jobs", [0, 1, 2])
@pytest.mark.parametrize("dims", [(3, 12, 8), (70, 40, 16)])
@pytest.mark.parametrize("has_se", [True, False])
def test_se_bnb_bwd(gpu, has_se, dims, jobs):
    """Composite check: u = bn_b(braw) [train stats] -> (SE gate) -> v ; L = sum(dv * v).  The kernel sees only the
    per-(n,c) sums; its coefficients must reproduce dL/dbraw, and the SE / BN parameter gradients.
    jobs: weight-gradient slab reductions riding on the launch (x3d_hip.h `reduce`): dw += the sum of its partial slabs, parts
    in ascending order -- compared with the fp64 sum and, bit for bit, with x3d_dw_slab_reduce on the same slabs."""
    ops = _ops()
    n, c, wd = dims
    gj = _gen(90 + jobs)
    red = []
    for parts, elems in [(251, 96 * 216), (7, 20 * 12)][:jobs]:
        slab = torch.randn((parts * elems,), generator=gj).to(gpu)
        red.append((slab, torch.full((elems,), 0.5, device=gpu), parts))
    T, H, W = 2, 3, 4
    P = T * H * W
    g_ = _gen(9)
    braw = torch.randn((n, c, T, H, W), generator=g_, dtype=torch.float64)
    dv = torch.randn((n, c, T, H, W), generator=g_, dtype=torch.float64)
    gamma = (1 + 0.2 * torch.randn(c, generator=g_)).double().requires_grad_(True)
    beta = (0.2 * torch.randn(c, generator=g_)).double().requires_grad_(True)
    w1 = (torch.randn((wd, c), generator=g_) * 0.3).double().requires_grad_(True)
    b1 = (torch.randn(wd, generator=g_) * 0.1).double().requires_grad_(True)
    w2 = (torch.randn((c, wd), generator=g_) * 0.3).double().requires_grad_(True)
    b2 = (torch.randn(c, generator=g_) * 0.1).double().requires_grad_(True)
    x_ = braw.clone().requires_grad_(True)
    mu = x_.mean((0, 2, 3, 4), keepdim=True)
    vr = x_.var((0, 2, 3, 4), unbiased=False, keepdim=True)
    inv = 1 / torch.sqrt(vr + 1e-5)
    u = (x_ - mu) * inv * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)
    if has_se:
        pooled = u.mean((2, 3, 4))
        h = F.relu(pooled @ w1.t() + b1)
        gate = torch.sigmoid(h @ w2.t() + b2)
        v = u * gate[:, :, None, None, None]
    else:
        v = u
    params = [x_, gamma, beta] + ([w1, b1, w2, b2] if has_se else [])
    grads = torch.autograd.grad((v * dv).sum(), params)
    # kernel inputs
    f32 = lambda t_: t_.detach().float().to(gpu).contiguous()
    bss = torch.stack([(gamma * inv.view(-1)).detach(), (beta - mu.view(-1) * gamma * inv.view(-1)).detach()], 1)
    bmi = torch.stack([mu.view(-1).detach(), inv.view(-1).detach()], 1)
    nc_sums = torch.stack([dv.sum((2, 3, 4)), (dv * braw).sum((2, 3, 4))], -1).to(gpu)
    pool_sums = braw.sum((2, 3, 4)).to(gpu)
    coef_nc = torch.empty((n, c, 4), device=gpu)
    dgam = torch.zeros(c, device=gpu)
    dbet = torch.zeros(c, device=gpu)
    kw = {}
    if has_se:
        kw = dict(w1=f32(w1), b1=f32(b1), w2=f32(w2), b2=f32(b2), gate=f32(gate), hidden=f32(h),
                  dw1=torch.zeros((wd, c), device=gpu), db1=torch.zeros(wd, device=gpu),
                  dw2=torch.zeros((c, wd), device=gpu), db2=torch.zeros(c, device=gpu),
                  scratch=torch.empty(n * (2 * c + wd), device=gpu))
    ops.se_bnb_bwd(nc_sums, pool_sums if has_se else None, P, f32(bss), f32(bmi), f32(gamma), dgam, dbet, coef_nc,
                   n, c, reduce=red, **kw)
    torch.cuda.synchronize()
    for slab, dwj, parts in red:
        ref = slab.double().view(parts, -1).sum(0) + 0.5
        report("slab sum", dwj, ref, 1e-6, 2e-6 * ref.abs().max().item())
        alone = torch.full_like(dwj, 0.5)
        ops.dw_slab_reduce([(slab, alone, parts)])
        torch.cuda.synchronize()
        assert torch.equal(alone, dwj), "the reduce slots of x3d_se_bnb_bwd and x3d_dw_slab_reduce add in the same order"
    cf = coef_nc.cpu().double()
    dB = cf[:, :, 0, None, None, None] * dv + cf[:, :, 1, None, None, None] * braw + cf[:, :, 2, None, None, None]
    report("dbraw", dB, grads[0], 2e-4, 2e-5)
    report("dgamma_b", dgam, grads[1], 2e-4, 2e-4)
    report("dbeta_b", dbet, grads[2], 2e-4, 2e-4)
    if has_se:
        report("dw1", kw["dw1"], grads[3], 2e-4, 2e-5)
        report("db1", kw["db1"], grads[4], 2e-4, 2e-5)
        report("dw2", kw["dw2"], grads[5], 2e-4, 2e-5)
        report("db2", kw["db2"], grads[6], 2e-4, 2e-5)


def test_head(gpu):
    ops = _ops()
    n, k, m1, m2 = 5, 54, 96, 40
    g_ = _gen(10)
    x = torch.randn((n, k), generator=g_)
    w1 = torch.randn((m1, k), generator=g_) * 0.2
    w2 = torch.randn((m2, m1), generator=g_) * 0.2
    b2 = torch.randn(m2, generator=g_) * 0.1
    mask = (torch.rand((n, m1), generator=g_) > 0.5).float()
    labels = torch.randint(0, m2, (n,), generator=g_, dtype=torch.int32)
    O = _oracle()
    xd = x.double().requires_grad_(True)
    w1d, w2d, b2d = [t.double().requires_grad_(True) for t in (w1, w2, b2)]
    h = F.relu(xd @ w1d.t())
    hm = h * mask.double() * 2.0
    logits = hm @ w2d.t() + b2d
    probs = torch.softmax(logits, -1)
    q = probs.clamp(1e-7, 1 - 1e-7)
    loss_rows = -torch.log(q.gather(1, labels.long().view(-1, 1)).squeeze(1)) + torch.log(q.sum(1))
    gx, gw1, gw2, gb2 = torch.autograd.grad(loss_rows.mean(), [xd, w1d, w2d, b2d])
    dev = lambda t_: t_.to(gpu)
    hg = torch.empty((n, m1), device=gpu)
    ops.dense_fwd(dev(x), dev(w1), None, hg, act=1)
    lg = torch.empty((n, m2), device=gpu)
    ops.dense_fwd(hg, dev(w2), dev(b2), lg, act=0, mask=dev(mask), mask_scale=2.0)
    pg = torch.empty((n, m2), device=gpu)
    lr = torch.empty(n, device=gpu)
    dl = torch.empty((n, m2), device=gpu)
    ops.softmax_xent(lg, dev(labels), pg, lr, dl, 1.0 / n)
    report("h", hg, h, 1e-5, 1e-5)
    report("logits", lg, logits, 1e-5, 1e-5)
    report("probs", pg, probs, 1e-5, 1e-7)
    report("loss_rows", lr, loss_rows, 1e-5, 1e-5)
    dh = torch.empty((n, m1), device=gpu)
    dw2 = torch.zeros((m2, m1), device=gpu)
    db2 = torch.zeros(m2, device=gpu)
    ops.dense_bwd(dl, None, 0, hg, dev(w2), dh, dw2, db2, mask=dev(mask), mask_scale=2.0)
    dx = torch.empty((n, k), device=gpu)
    dw1 = torch.zeros((m1, k), device=gpu)
    ops.dense_bwd(dh, hg, 1, dev(x), dev(w1), dx, dw1, None)
    report("dw2", dw2, gw2, 1e-4, 1e-6)
    report("db2", db2, gb2, 1e-4, 1e-6)
    report("dw1", dw1, gw1, 1e-4, 1e-6)
    report("dx", dx, gx, 1e-4, 1e-6)
    # clipped regime: a dominant logit pushes p to the 1e-7 clip
    big = torch.zeros((2, m2))
    big[0, 3] = 40.0
    big[1, 5] = 40.0
    lab2 = torch.tensor([3, 7], dtype=torch.int32)
    bd = big.double().requires_grad_(True)
    pr = torch.softmax(bd, -1)
    qq = pr.clamp(1e-7, 1 - 1e-7)
    ll = -torch.log(qq.gather(1, lab2.long().view(-1, 1)).squeeze(1)) + torch.log(qq.sum(1))
    (gbig,) = torch.autograd.grad(ll.sum(), [bd])
    p2 = torch.empty((2, m2), device=gpu)
    l2 = torch.empty(2, device=gpu)
    d2 = torch.empty((2, m2), device=gpu)
    ops.softmax_xent(dev(big), dev(lab2), p2, l2, d2, 1.0)
    report("loss_clipped", l2, ll, 1e-4, 1e-4)
    report("dlogits_clipped", d2, gbig, 1e-3, 1e-6)
    # view mean
    out = torch.empty((1, m2), device=gpu)
    ops.view_mean(p2, out, 2)
    report("view_mean", out, p2.cpu().double().mean(0, keepdim=True), 1e-6, 1e-8)


def test_sgd_and_layout(gpu):
    ops, O = _ops(), _oracle()
    g_ = _gen(11)
    nel = 1000
    w = torch.randn(nel, generator=g_)
    v = torch.randn(nel, generator=g_) * 0.1
    g = torch.randn(nel, generator=g_)
    mask = (torch.rand(nel, generator=g_) > 0.5).to(torch.uint8)
    wg, vg = w.to(gpu), v.to(gpu)
    ops.sgd_nesterov(wg, vg, g.to(gpu), mask.to(gpu), 0.1, 0.9, 5e-5, 0.5)
    gg = g.double() * 0.5 + 2 * 5e-5 * w.double() * mask.double()
    vref = 0.9 * v.double() - 0.1 * gg
    wref = w.double() + 0.9 * vref - 0.1 * gg
    report("v", vg, vref, 1e-6, 1e-7)
    report("w", wg, wref, 1e-6, 1e-7)
    acc = torch.zeros(1, dtype=torch.float64, device=gpu)
    ops.l2_sumsq(w.to(gpu), mask.to(gpu), acc)
    report("l2", acc, ((w.double() ** 2) * mask.double()).sum().view(1), 1e-6, 1e-6)
    x = torch.randn((2, 3, 5, 7, 3), generator=g_)
    for dt in DTYPES:
        dst = torch.empty((2, 3, 3, 5, 7), dtype=dt, device=gpu)
        ops.nthwc_to_ncthw(x.to(gpu), dst)
        report("layout", dst, x.permute(0, 4, 1, 2, 3).to(dt), 0, 0)
    # the 8-points-per-thread form (3 channels, P % 8 == 0), every source / destination storage type pair the model uses
    x8 = torch.randn((3, 2, 6, 12, 3), generator=g_)
    for sdt in DTYPES:
        for dt in DTYPES:
            if sdt != torch.float32 and dt not in (sdt, torch.float32):
                continue
            src = x8.to(sdt).to(gpu)
            dst = torch.empty((3, 3, 2, 6, 12), dtype=dt, device=gpu)
            ops.nthwc_to_ncthw(src, dst)
            report(f"layout8 {sdt}->{dt}", dst, x8.to(sdt).permute(0, 4, 1, 2, 3).to(dt), 0, 0)


# --------------------------------------------------------------------------------------------------
# BatchNorm finalize folded into the consumer (x3d_bn_fold): the same bits as x3d_bn_finalize + the plain kernel
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(3, 5, 4, 14, 14, 1), (2, 3, 3, 16, 16, 2), (2, 4, 5, 28, 28, 1), (1, 2, 2, 56, 56, 1)])
def test_bn_fold_dw3d_fwd(gpu, dtype, shape):
    ops = _ops()
    n, c, t, h, w, stride = shape
    g = _gen(21)
    x, _ = rnd((n, c, t, h, w), dtype, g)
    x = x.to(gpu)
    wt = (torch.randn((c, 3, 3, 3), generator=g) * 0.3).to(gpu)
    xs = x.float().double()
    stats = torch.stack([xs.sum((0, 2, 3, 4)), (xs * xs).sum((0, 2, 3, 4))], 1).contiguous()
    gamma = (1 + 0.3 * torch.randn(c, generator=g)).to(gpu)
    beta = (0.3 * torch.randn(c, generator=g)).to(gpu)
    count = n * t * h * w

    def fresh():
        return (torch.full((c,), 0.25, device=gpu), torch.full((c,), 1.5, device=gpu), torch.zeros((c, 2), device=gpu),
                torch.zeros((c, 2), device=gpu))
    mm0, mv0, ss0, mi0 = fresh()
    ops.bn_finalize(stats, count, gamma, beta, mm0, mv0, 1e-5, 0.9, 1, ss0, mi0)
    st0 = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    y0 = ops.dw3d_fwd(x, wt, stride, in_ss=ss0, in_act=1, stats=st0)
    # the folded form always runs a vector kernel; where the plain launch goes to the matrix cores (dw_mx.hip: operands
    # rounded to the storage type) the outputs are compared at the matrix-core tolerance instead of bit for bit (the
    # library reads its A/B switches once per process, so X3D_DW_MX cannot be flipped here)
    on_mx = _dw_on_matrix_cores(S.dw_fwd_struct(shape, dtype))
    mm1, mv1, ss1, mi1 = fresh()
    fold = ops.bn_fold(stats, count, gamma, beta, mm1, mv1, 1e-5, 0.9, 1, ss1, mi1)
    st1 = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    y1 = ops.dw3d_fwd(x, wt, stride, in_act=1, stats=st1, in_bn=fold)
    torch.cuda.synchronize()
    if on_mx:
        # (operand rounding of 27 products on one side only: 1 % of the tensor's scale; the bit-exact form of this check
        # runs on the shapes that stay on the vector kernels)
        report("bn_fold dw3d_fwd (matrix-core plain launch)", y0, y1.float(), 0, 1e-2 * y1.float().abs().max().item())
    else:
        assert torch.equal(y0, y1)
    for a_, b_ in ((ss0, ss1), (mi0, mi1), (mm0, mm1), (mv0, mv1)):
        assert torch.equal(a_, b_)
    assert ss1.abs().sum().item() > 0 and not torch.equal(mm1, torch.full((c,), 0.25, device=gpu))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 6, 3 * 14 * 14, True), (3, 4, 5 * 7 * 7, False), (1, 5, 1000, None)])
def test_bn_fold_tail_fwd(gpu, dtype, shape):
    ops = _ops()
    n, c, pts, conv_shortcut = shape
    g = _gen(22)
    craw, _ = rnd((n, c, 1, 1, pts), dtype, g)
    craw = craw.to(gpu)
    sh = None
    if conv_shortcut is not None:
        sh, _ = rnd((n, c, 1, 1, pts), dtype, g)
        sh = sh.to(gpu)

    def stats_of(v):
        d = v.float().double()
        return torch.stack([d.sum((0, 2, 3, 4)), (d * d).sum((0, 2, 3, 4))], 1).contiguous()

    def bn_set():
        return dict(gamma=(1 + 0.3 * torch.randn(c, generator=_gen(5))).to(gpu), beta=(0.3 * torch.randn(c, generator=_gen(6))).to(gpu),
                    mm=torch.full((c,), -0.5, device=gpu), mv=torch.full((c,), 2.0, device=gpu),
                    ss=torch.zeros((c, 2), device=gpu), mi=torch.zeros((c, 2), device=gpu))
    count = n * pts
    ref_c, ref_r, new_c, new_r = bn_set(), bn_set(), bn_set(), bn_set()
    st_c = stats_of(craw)
    ops.bn_finalize(st_c, count, ref_c["gamma"], ref_c["beta"], ref_c["mm"], ref_c["mv"], 1e-5, 0.9, 1, ref_c["ss"], ref_c["mi"])
    fc = ops.bn_fold(st_c, count, new_c["gamma"], new_c["beta"], new_c["mm"], new_c["mv"], 1e-5, 0.9, 1, new_c["ss"], new_c["mi"])
    fr, r_ss = None, None
    if conv_shortcut:
        st_r = stats_of(sh)
        ops.bn_finalize(st_r, count, ref_r["gamma"], ref_r["beta"], ref_r["mm"], ref_r["mv"], 1e-5, 0.9, 1, ref_r["ss"], ref_r["mi"])
        fr = ops.bn_fold(st_r, count, new_r["gamma"], new_r["beta"], new_r["mm"], new_r["mv"], 1e-5, 0.9, 1, new_r["ss"], new_r["mi"])
        r_ss = ref_r["ss"]
    y0 = ops.tail_fwd(craw, ref_c["ss"], sh, r_ss, torch.empty_like(craw))
    y1 = ops.tail_fwd_bn(craw, fc, sh, fr, torch.empty_like(craw))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    for k in ("ss", "mi", "mm", "mv"):
        assert torch.equal(ref_c[k], new_c[k]), k
        if conv_shortcut:
            assert torch.equal(ref_r[k], new_r[k]), k


def test_adam_finite_and_label_range(gpu):
    ops = _ops()
    from x3d_tf_amd import hip
    g_ = _gen(12)
    nel = 1000
    w = torch.randn(nel, generator=g_)
    m1 = torch.randn(nel, generator=g_) * 0.1
    v2 = torch.rand(nel, generator=g_) * 0.01
    g = torch.randn(nel, generator=g_)
    mask = (torch.rand(nel, generator=g_) > 0.5).to(torch.uint8)
    wg, mg, vg, gg_, maskg = w.to(gpu), m1.to(gpu), v2.to(gpu), g.to(gpu), mask.to(gpu)
    step, lr, b1, b2, eps, wd, gs = 3, 0.01, 0.9, 0.999, 1e-7, 5e-5, 0.5
    hip.call("x3d_adam", wg.data_ptr(), mg.data_ptr(), vg.data_ptr(), gg_.data_ptr(), maskg.data_ptr(), lr, b1, b2,
             eps, wd, gs, step, nel)
    torch.cuda.synchronize()
    gg = g.double() * gs + 2 * wd * w.double() * mask.double()
    mref = b1 * m1.double() + (1 - b1) * gg
    vref = b2 * v2.double() + (1 - b2) * gg * gg
    wref = w.double() - lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step) * mref / (vref.sqrt() + eps)
    # (1 - beta) is formed in fp32 as Keras does: 1 - 0.999f = 0.00100005 (5e-5 relative on the second-moment increment)
    report("adam m", mg, mref, 1e-5, 1e-7)
    report("adam v", vg, vref, 1e-4, 1e-9)
    report("adam w", wg, wref, 1e-4, 1e-6)
    # x3d_all_finite
    flag = torch.ones(1, dtype=torch.int32, device=gpu)
    big = torch.randn(100003, generator=g_).to(gpu)
    hip.call("x3d_all_finite", big.data_ptr(), big.numel(), flag.data_ptr())
    assert flag.item() == 1
    for bad in (float("inf"), float("-inf"), float("nan")):
        big2 = big.clone()
        big2[77777] = bad
        flag.fill_(1)
        hip.call("x3d_all_finite", big2.data_ptr(), big2.numel(), flag.data_ptr())
        assert flag.item() == 0
    # a label outside [0, M): NaN loss row, zero gradient row, no out-of-bounds read; the other rows are untouched
    logits = torch.randn(3, 17, generator=g_).to(gpu)
    lab = torch.tensor([2, 99, -1], dtype=torch.int32, device=gpu)
    probs, rows, dl = torch.empty(3, 17, device=gpu), torch.empty(3, device=gpu), torch.full((3, 17), 7.0, device=gpu)
    ops.softmax_xent(logits, lab, probs, rows, dl, 1.0)
    assert torch.isfinite(rows[0]) and torch.isnan(rows[1]) and torch.isnan(rows[2])
    assert dl[1].abs().sum().item() == 0 and dl[2].abs().sum().item() == 0 and dl[0].abs().sum().item() > 0
    report("probs", probs, torch.softmax(logits.double().cpu(), -1), 1e-5, 1e-7)


@pytest.mark.gpu
def test_pw_f32_tensors_over_2gb(gpu):
    """fp32 storage, tensors of more than 2 GB (52 x 54 x 16x112x112 floats = 2.25 GB).  The pipelined fp32 kernels address a whole tensor with
    32-bit buffer offsets and hand such launches to the resident-weights kernels behind the same entry points (pw_gemm_f32r.h /
    pw_wgrad_f32r.h; per-sample buffer resources were measured: the limit goes, the X3D-S step pays 1.7 %) -- this is the case that keeps
    those kernels under test.  Size-independent properties: a sample's forward output / data gradient does not depend on the batch it is in
    (bit-identical to a launch of the last two samples alone), the statistics are the sums of the stored output, the weight gradient is
    the sum over sub-batches."""
    ops = _ops()
    from x3d_tf_amd import hip
    n, cin, cout, t, h, w = 52, 24, 54, 16, 112, 112
    names = [hip.kernel_name(st) for st in (S.pw_fwd_struct((n, cin, cout, t, h, w, 1, None), S.F32, False),
                                            S.pw_dgrad_struct((n, cin, cout, t, h, w), "add", S.F32, False),
                                            S.pw_wgrad_struct((n, cin, cout, t, h, w, 1, None), S.F32))]
    assert [k.split("<")[0] for k in names] == ["pw_f32r_kernel", "pw_f32r_kernel", "pw_wgrad_f32r_kernel"], names
    g_ = torch.Generator(device=gpu).manual_seed(11)
    x = torch.randn((n, cin, t, h, w), generator=g_, device=gpu)
    wt = torch.randn((cout, cin), generator=g_, device=gpu) * 0.2
    # forward
    stats = torch.zeros((cout, 2), dtype=torch.float64, device=gpu)
    y = ops.pw_fwd(x, wt, stats=stats)
    assert y.numel() * 4 > (1 << 31)
    y2 = ops.pw_fwd(x[-2:].contiguous(), wt, stats=torch.zeros((cout, 2), dtype=torch.float64, device=gpu))
    torch.cuda.synchronize()
    assert torch.equal(y[-2:], y2), "forward: the last samples of the 2 GB launch differ from the same samples alone"
    ref = torch.einsum("oc,cp->op", wt.double(), x[-1].reshape(cin, -1).double())
    assert (y[-1].reshape(cout, -1).double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    s_ref = torch.stack([torch.stack([y[:, c].double().sum(), (y[:, c].double() ** 2).sum()]) for c in range(cout)])
    assert (stats - s_ref).abs().max().item() <= 2e-4 * s_ref.abs().max().item()
    # data gradient (dY = A g + B yraw + C, dx = W^T dY + add) and weight gradient on the same g / yraw
    g = y                                                   # any 2 GB tensor of the right shape
    yraw = torch.randn(y.shape, generator=g_, device=gpu)
    coef = torch.randn((cout, 4), generator=g_, device=gpu) * 0.5
    add = torch.randn(x.shape, generator=g_, device=gpu)
    dx = torch.empty_like(x)
    ops.pw_dgrad(g, yraw, coef, wt, dx, ops.EPI_ADD, add=add)
    dx2 = torch.empty_like(x[-2:])
    ops.pw_dgrad(g[-2:].contiguous(), yraw[-2:].contiguous(), coef, wt, dx2, ops.EPI_ADD, add=add[-2:].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(dx[-2:], dx2), "data gradient: the last samples of the 2 GB launch differ from the same samples alone"
    dw = torch.zeros((cout, cin), device=gpu)
    ops.pw_wgrad(g, yraw, coef, x, dw)
    dws = torch.zeros((cout, cin), device=gpu)
    for i in range(0, n, 13):
        ops.pw_wgrad(g[i:i + 13].contiguous(), yraw[i:i + 13].contiguous(), coef, x[i:i + 13].contiguous(), dws)
    torch.cuda.synchronize()
    assert (dw - dws).abs().max().item() <= 1e-4 * dws.abs().max().item(), "weight gradient: 2 GB launch vs the sum over sub-batches"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(32, 96, 216, 13, 10, 10), (32, 432, 192, 13, 5, 5), (32, 24, 54, 13, 40, 40)])   # N, Cin, Cout, T, H, W
def test_pw_f32_full_size_fp64(gpu, shape):
    """The pipelined fp32 pointwise kernels at BASELINE config 2's real layer sizes (X3D-S, 32 clips of 13x160x160: stages 4, 5 and 2)
    against fp64 arithmetic on the GPU -- independent of the CPU oracle: forward with statistics, data gradient with the residual add,
    weight gradient.  Limit 2e-5 of the largest value (the kernel tests' fp32 tol_gemm); measured on an MI355X: forward 2.3e-7 ... 1.0e-6,
    statistics 1.8e-9 ... 3.5e-8, data gradient 2.9e-7 ... 5.7e-7, weight gradient (10 400 ... 665 600 points in partial fp32 tiles) 3.5e-7 ... 6.9e-7."""
    ops = _ops()
    n, cin, cout, t, h, w = shape
    g_ = torch.Generator(device=gpu).manual_seed(21)
    x = torch.randn((n, cin, t, h, w), generator=g_, device=gpu)
    wt = torch.randn((cout, cin), generator=g_, device=gpu) * (2.0 / cin) ** 0.5
    stats = torch.zeros((cout, 2), dtype=torch.float64, device=gpu)
    y = ops.pw_fwd(x, wt, stats=stats)
    xd, wd = x.double().reshape(n, cin, -1), wt.double()
    ref = torch.einsum("oc,ncp->nop", wd, xd)
    torch.cuda.synchronize()
    err = (y.double().reshape(n, cout, -1) - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5, f"forward {err:.2e}"
    s_ref = torch.stack([ref.sum((0, 2)), (ref ** 2).sum((0, 2))], 1)
    assert ((stats - s_ref).abs().max() / s_ref.abs().max()).item() <= 2e-5, "statistics"
    gy = torch.randn((n, cout, t, h, w), generator=g_, device=gpu)
    yraw = torch.randn((n, cout, t, h, w), generator=g_, device=gpu)
    coef = torch.randn((cout, 4), generator=g_, device=gpu) * 0.5
    add = torch.randn(x.shape, generator=g_, device=gpu)
    dx = torch.empty_like(x)
    ops.pw_dgrad(gy, yraw, coef, wt, dx, ops.EPI_ADD, add=add)
    cd = coef.double()
    dy = cd[:, 0].view(1, -1, 1) * gy.double().reshape(n, cout, -1) + cd[:, 1].view(1, -1, 1) * yraw.double().reshape(n, cout, -1) + cd[:, 2].view(1, -1, 1)
    ref_dx = torch.einsum("oc,nop->ncp", wd, dy) + add.double().reshape(n, cin, -1)
    torch.cuda.synchronize()
    err = (dx.double().reshape(n, cin, -1) - ref_dx).abs().max().item() / ref_dx.abs().max().item()
    assert err <= 2e-5, f"data gradient {err:.2e}"
    dw = torch.zeros((cout, cin), device=gpu)
    ops.pw_wgrad(gy, yraw, coef, x, dw)
    ref_dw = torch.einsum("nop,ncp->oc", dy, xd)
    torch.cuda.synchronize()
    err = (dw.double() - ref_dw).abs().max().item() / ref_dw.abs().max().item()
    assert err <= 2e-5, f"weight gradient {err:.2e}"
