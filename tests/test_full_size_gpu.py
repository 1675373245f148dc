"""Single layers at BASELINE config 3's FULL size (X3D-M, 64 clips of 16 x 224^2, bf16 / fp16 storage) against fp64 on the GPU.

The kernel-level parity cases (tests/test_kernels_gpu.py) run shapes a CPU finishes in seconds: at most 150 k points per
channel.  Three kernels of the headline step reduce over 3.2 - 12.8 M points in fp32 (MFMA accumulators, per-thread partial
sums, fp32 atomics), one of them -- the recomputed-output `a` backward, pw_bwd_rc.hip -- through a CANCELLING expression
(dW = diag(A)(sum g x^T) + diag(B) W (sum x x^T) + C (sum x)^T with a post-ReLU x of positive mean: the C term removes the mean
part of the first).  "Finite and linear in the upstream gradient" (test_full_size_plan_properties) passes for any amount of
cancellation error, so here the same launches are compared with the fp64 definition at the real reduction length.

The fp64 side is torch on the GPU (einsum / shifted slices), TEST SIDE ONLY, chunked over samples; it restates the same
definitions the small cases check against the CPU oracle (test_pw_bwd_rc, test_pw_wgrad, test_dw3d_bwd), with the
BatchNorm-backward coefficients DERIVED from the data (sum dY = 0 and sum dY * yhat = 0 per channel, as in a real step) instead
of drawn at random -- that is what makes the moment sums cancel.
"""
import pytest
import torch

from tests.util import round_to, tol_gemm, tol_store

pytestmark = pytest.mark.gpu
HALF = [torch.bfloat16, torch.float16]


def _wtol(dtype):
    return 1e-3 if dtype == torch.bfloat16 else 4e-4      # as tests/test_kernels_gpu.py::_wtol


def _bn_bwd_coef(g, y, gamma):
    """[C][4] fp32 coefficients (A, B, C, 0) of dY = A g + B y + C for training-mode BatchNorm over (N, T, H, W), from the
    tensors themselves: fp64 statistics, as x3d_bn_finalize / x3d_bn_bwd_finalize produce them."""
    m = y.shape[0] * y.shape[2] * y.shape[3] * y.shape[4]
    mean = y.mean((0, 2, 3, 4))
    var = (y * y).mean((0, 2, 3, 4)) - mean * mean
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    dbe = g.sum((0, 2, 3, 4))
    dga = ((g * y).sum((0, 2, 3, 4)) - mean * dbe) * invstd
    k1 = gamma * invstd
    b = -k1 * invstd * dga / m
    c = -k1 * dbe / m - b * mean
    return torch.stack([k1, b, c, torch.zeros_like(c)], 1).float()


def _check(name, got, ref, rtol, atol):
    got, ref = got.double(), ref.double()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    assert not bad.any(), (f"{name}: {int(bad.sum())}/{bad.numel()} out of tolerance (rtol {rtol}, atol {atol:.3e}); max err "
                           f"{err.max().item():.3e}, max |ref| {ref.abs().max().item():.3e}")
    return (err.max() / (ref.abs().max() + 1e-300)).item()


@pytest.mark.parametrize("dtype", HALF)
def test_block0_a_backward_recomputed_output_full_size(gpu, dtype):
    """x3d_pw_bwd, rc form, block 0 of X3D-M at batch 64: 24 <-> 54 channels on 16 x 112 x 112 points (12.8 M per channel), the
    strided shortcut-gradient add and the stem-fold tail, post-ReLU x with mean/std ~ 1.  dx against the fold with its panel
    operands rounded (tol_gemm), dW against the DEFINITION dY x^T in fp64 (_wtol: 1e-3 / 4e-4 of its maximum)."""
    from x3d_tf_amd import ops
    n, cin, cout, t, h, w = 64, 24, 54, 16, 112, 112
    g_ = torch.Generator(device=gpu)
    g_.manual_seed(41)
    rn = lambda *s: torch.randn(*s, generator=g_, device=gpu, dtype=torch.float32)
    t_raw = (rn(n, cin, t, h, w) + 0.6).to(dtype)                       # the stem's conv_t output; x = relu(.) = y0
    x = torch.relu(t_raw.float()).to(dtype)
    wt = rn(cout, cin) * 0.2
    gamma = 1 + 0.3 * rn(cout)
    beta = 0.3 * rn(cout)
    wr = round_to(wt.cpu(), dtype).to(gpu)
    # upstream gradient = what the depthwise backward emits: masked by the ReLU behind bn_a
    g = torch.empty((n, cout, t, h, w), dtype=dtype, device=gpu)
    CH = 8
    s1 = torch.zeros(cout, dtype=torch.float64, device=gpu)
    s2 = torch.zeros(cout, dtype=torch.float64, device=gpu)
    for i in range(0, n, CH):        # statistics of y = Wr x
        y = torch.einsum("oc,ncthw->nothw", wr, x[i:i + CH].double())
        s1 += y.sum((0, 2, 3, 4))
        s2 += (y * y).sum((0, 2, 3, 4))
        del y
    m = n * t * h * w
    mean = s1 / m
    invstd = 1.0 / torch.sqrt(s2 / m - mean * mean + 1e-5)
    dbe = torch.zeros(cout, dtype=torch.float64, device=gpu)
    dgy = torch.zeros(cout, dtype=torch.float64, device=gpu)
    for i in range(0, n, CH):
        y = torch.einsum("oc,ncthw->nothw", wr, x[i:i + CH].double())
        z = (y - mean.view(1, -1, 1, 1, 1)) * (gamma.double() * invstd).view(1, -1, 1, 1, 1) + beta.double().view(1, -1, 1, 1, 1)
        gi = (rn(CH, cout, t, h, w) * (z > 0)).to(dtype)
        g[i:i + CH] = gi
        dbe += gi.double().sum((0, 2, 3, 4))
        dgy += (gi.double() * y).sum((0, 2, 3, 4))
        del y, z, gi
    dga = (dgy - mean * dbe) * invstd
    k1 = gamma.double() * invstd
    bb = -k1 * invstd * dga / m
    cc = -k1 * dbe / m - bb * mean
    coef = torch.stack([k1, bb, cc, torch.zeros_like(cc)], 1).float()
    add = rn(n, cin, t, h // 2, w // 2).to(dtype)                        # the strided shortcut conv's data gradient
    dx = torch.empty((n, cin, t, h, w), dtype=dtype, device=gpu)
    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)
    sums_c = torch.zeros((cin, 2), dtype=torch.float64, device=gpu)
    ok = ops.pw_bwd_rc(g, x, wt, coef, dx, dw, ops.EPI_ADD_STRIDED, add, tail_c=t_raw, tail_sums_c=sums_c)
    torch.cuda.synchronize()
    assert ok, "the recomputed-output form should cover block 0 of X3D-M"
    # ---- fp64 definition, chunked
    c = coef.double()
    w1 = round_to((wr.cpu() * c[:, 0:1].cpu()).float(), dtype).to(gpu)
    mm = round_to(torch.einsum("oc,o,od->cd", wr, c[:, 1], wr).float().cpu(), dtype).to(gpu)
    c0 = (wr * c[:, 2:3]).sum(0)
    dw_ref = torch.zeros((cout, cin), dtype=torch.float64, device=gpu)
    rt, at = tol_gemm(dtype)
    worst_dx, scale = 0.0, 0.0
    ts1 = torch.zeros(cin, dtype=torch.float64, device=gpu)
    ts2 = torch.zeros(cin, dtype=torch.float64, device=gpu)
    refs = []
    for i in range(0, n, CH):
        xd, gd = x[i:i + CH].double(), g[i:i + CH].double()
        y = torch.einsum("oc,ncthw->nothw", wr, xd)
        dy = c[:, 0].view(1, -1, 1, 1, 1) * gd + c[:, 1].view(1, -1, 1, 1, 1) * y + c[:, 2].view(1, -1, 1, 1, 1)
        dw_ref += torch.einsum("nothw,ncthw->oc", dy, xd)
        del y, dy
        ref = (torch.einsum("oc,nothw->ncthw", w1, gd) + torch.einsum("cd,ndthw->ncthw", mm, xd) + c0.view(1, -1, 1, 1, 1))
        ref[:, :, :, ::2, ::2] += add[i:i + CH].double()
        ref = ref * (xd > 0)
        scale = max(scale, ref.abs().max().item())
        refs.append((i, ref))
        dxs = dx[i:i + CH].double()
        ts1 += dxs.sum((0, 2, 3, 4))
        ts2 += (dxs * t_raw[i:i + CH].double()).sum((0, 2, 3, 4))
        if len(refs) == 2 or i + CH >= n:      # (two chunks at a time in memory)
            for j, rf in refs:
                worst_dx = max(worst_dx, _check(f"dx[{j}:{j + CH}]", dx[j:j + CH], rf, rt, at * max(scale, 1e-30)))
            refs = []
    # (measured: 2.5e-6 / 1.9e-6 of the maximum -- profiles/r05_full_size_fp64_checks.txt -- although the sum cancels to 1e-3 of
    # its terms; the kernel tests' limit for this quantity is 1e-3 / 4e-4, here a tenth of it)
    tol = 0.1 * _wtol(dtype)
    e_dw = _check("dw", dw.double() - 0.5, dw_ref, tol, tol * dw_ref.abs().max().item())
    st = 3e-3 if dtype == torch.bfloat16 else 5e-4
    sref = torch.stack([ts1, ts2], 1)
    _check("tail_sums_c", sums_c, sref, 10 * st, 10 * st * max(1.0, sref.abs().max().item()))
    print(f"full-size block-0 `a` backward {dtype}: dx err {worst_dx:.2e} of max, dW err {e_dw:.2e} of max (limit {tol:.0e}); "
          f"|dW| max {dw_ref.abs().max().item():.3e}, term scale {(c[:, 0].abs().max() * g.double().abs().mean() * x.double().mean() * m).item():.3e}")


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("which", ["c", "a"])
def test_stage5_weight_gradient_full_size(gpu, dtype, which):
    """x3d_pw_wgrad of stage 5 at batch 64 (50 176 points per channel, 7 x 7 planes): `c` conv 192 x 432 with the
    BN_b * gate -> swish prologue, `a` conv 432 x 192 on the block input; coefficients derived from the data."""
    from x3d_tf_amd import ops
    n, t, h, w = 64, 16, 7, 7
    cin, cout = (432, 192) if which == "c" else (192, 432)
    g_ = torch.Generator(device=gpu)
    g_.manual_seed(43)
    rn = lambda *s: torch.randn(*s, generator=g_, device=gpu, dtype=torch.float32)
    x = rn(n, cin, t, h, w).to(dtype)
    if which == "a":
        x = torch.relu(x.float() + 0.3).to(dtype)
    yraw = (rn(n, cout, t, h, w) * 0.8 + 0.2).to(dtype)
    g = rn(n, cout, t, h, w).to(dtype)
    coef = _bn_bwd_coef(g.double(), yraw.double(), (1 + 0.3 * rn(cout)).double())
    ss = gate = None
    act = 0
    xin = x.double()
    if which == "c":
        ss = torch.stack([1 + 0.3 * rn(cin), 0.3 * rn(cin)], 1)
        gate = torch.rand((n, cin), generator=g_, device=gpu)
        act = 2
        u = (x.float() * ss[:, 0].view(1, -1, 1, 1, 1) + ss[:, 1].view(1, -1, 1, 1, 1)) * gate[:, :, None, None, None]
        xin = (u * torch.sigmoid(u)).to(dtype).double()
    cf = coef.float()
    dy = (cf[:, 0].view(1, -1, 1, 1, 1) * g.float() + cf[:, 1].view(1, -1, 1, 1, 1) * yraw.float() + cf[:, 2].view(1, -1, 1, 1, 1)).to(dtype).double()
    ref = torch.einsum("nothw,ncthw->oc", dy, xin)
    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=gpu)
    ops.pw_wgrad(g, yraw, coef, x, dw, in_ss=ss, in_gate=gate, in_act=act)
    torch.cuda.synchronize()
    tol = 0.3 * _wtol(dtype)        # (measured 6.3e-5 / 1.4e-5 of the maximum)
    e = _check("dw", dw.double() - 0.5, ref, tol, tol * ref.abs().max().item())
    print(f"full-size stage-5 `{which}` weight gradient {dtype}: err {e:.2e} of max (limit {tol:.0e})")


@pytest.mark.parametrize("dtype", HALF)
def test_depthwise_56_backward_full_size(gpu, dtype):
    """x3d_dw3d_bwd of the stage-2 stride-1 layers at batch 64: 54 channels of 16 x 56 x 56 (3.2 M points per channel and
    weight tap), fused data + weight gradient with the BN_b / SE backward on load and the ReLU mask + BN_a sums in the
    epilogue, against the fp64 stencil (shifted slices)."""
    from x3d_tf_amd import hip, ops
    from tests import shapes as S
    n, c, t, h, w = 64, 54, 16, 56, 56
    g_ = torch.Generator(device=gpu)
    g_.manual_seed(47)
    rn = lambda *s: torch.randn(*s, generator=g_, device=gpu, dtype=torch.float32)
    araw = rn(n, c, t, h, w).to(dtype)
    dv = rn(n, c, t, h, w).to(dtype)
    braw = rn(n, c, t, h, w).to(dtype)
    coef = rn(n, c, 4) * 0.5
    wt = rn(c, 3, 3, 3) * 0.3
    ss = torch.stack([1 + 0.3 * rn(c), 0.3 * rn(c)], 1)
    ga = torch.empty((n, c, t, h, w), dtype=dtype, device=gpu)
    a_sums = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    dw = torch.full((c, 27), 0.25, dtype=torch.float32, device=gpu)
    ops.dw3d_bwd(dv, braw, coef, araw, ss, wt.view(c, 27), ga, a_sums, dw, 1)
    torch.cuda.synchronize()
    mx = "_mx" in hip.kernel_name(S.dw_bwd_struct((n, c, t, h, w, 1), dtype))
    wd = round_to(wt.cpu(), dtype).to(gpu) if mx else wt.double()
    dw_ref = torch.zeros((c, 3, 3, 3), dtype=torch.float64, device=gpu)
    worst, CH = 0.0, 8
    s1 = torch.zeros(c, dtype=torch.float64, device=gpu)
    s2 = torch.zeros(c, dtype=torch.float64, device=gpu)
    rt, at = tol_gemm(dtype) if mx else tol_store(dtype)
    for i in range(0, n, CH):
        cd = coef[i:i + CH].double()
        dB = cd[:, :, 0, None, None, None] * dv[i:i + CH].double() + cd[:, :, 1, None, None, None] * braw[i:i + CH].double() + cd[:, :, 2, None, None, None]
        z = araw[i:i + CH].double() * ss[:, 0].double().view(1, -1, 1, 1, 1) + ss[:, 1].double().view(1, -1, 1, 1, 1)
        act = torch.relu(z)
        if mx:
            dB = dB.float().to(dtype).double()
            act = act.float().to(dtype).double()
        pa = torch.nn.functional.pad(act, (1, 1, 1, 1, 1, 1))
        pb = torch.nn.functional.pad(dB, (1, 1, 1, 1, 1, 1))
        dA = torch.zeros_like(act)
        for kt in range(3):
            for kh in range(3):
                for kw in range(3):
                    # out[p] = sum_k w[k] a[p + k - 1]  =>  dA[q] = sum_k w[k] dB[q - k + 1],  dW[k] = sum_p dB[p] a[p + k - 1]
                    dA += wd[:, kt, kh, kw].view(1, -1, 1, 1, 1) * pb[:, :, 2 - kt:2 - kt + t, 2 - kh:2 - kh + h, 2 - kw:2 - kw + w]
                    dw_ref[:, kt, kh, kw] += (dB * pa[:, :, kt:kt + t, kh:kh + h, kw:kw + w]).sum((0, 2, 3, 4))
        ref = dA * (z > 0)
        worst = max(worst, _check(f"ga[{i}:{i + CH}]", ga[i:i + CH], ref, rt, at * ref.abs().max().item()))
        gs = ga[i:i + CH].double()
        s1 += gs.sum((0, 2, 3, 4))
        s2 += (gs * araw[i:i + CH].double()).sum((0, 2, 3, 4))
        del dB, z, act, pa, pb, dA, ref, gs
    wtol = _wtol(dtype) if mx else 2e-5        # (vector kernel, fp32 products: measured 5.6e-7 of the maximum)
    e = _check("dw", dw.double().view(c, 3, 3, 3) - 0.25, dw_ref, wtol, wtol * dw_ref.abs().max().item())
    st = 3e-3 if dtype == torch.bfloat16 else 5e-4
    sref = torch.stack([s1, s2], 1)
    _check("a_sums", a_sums, sref, 10 * st, 10 * st * max(1.0, sref.abs().max().item()))
    print(f"full-size 56^2 depthwise backward {dtype} ({'matrix-core' if mx else 'vector'} kernel): ga err {worst:.2e} of max, dW err {e:.2e} of max (limit {wtol:.0e})")


@pytest.mark.parametrize("dtype", HALF)
def test_depthwise_112_stride2_backward_full_size(gpu, dtype):
    """x3d_dw3d_bwd of the first block's stride-2 layer at batch 64: 54 channels of 16 x 112 x 112 -> 56 x 56 (the launch with the
    largest total of the train step; dw3d_bwd_s2r_kernel for 16-bit storage), against the fp64 stencil written as strided slices
    (TF-SAME for an even extent at stride 2: no pad in front, one behind)."""
    from x3d_tf_amd import hip, ops
    from tests import shapes as S
    n, c, t, h, w = 64, 54, 16, 112, 112
    ho, wo = h // 2, w // 2
    g_ = torch.Generator(device=gpu)
    g_.manual_seed(53)
    rn = lambda *s: torch.randn(*s, generator=g_, device=gpu, dtype=torch.float32)
    araw = rn(n, c, t, h, w).to(dtype)
    dv = rn(n, c, t, ho, wo).to(dtype)
    braw = rn(n, c, t, ho, wo).to(dtype)
    coef = rn(n, c, 4) * 0.5
    wt = rn(c, 3, 3, 3) * 0.3
    ss = torch.stack([1 + 0.3 * rn(c), 0.3 * rn(c)], 1)
    ga = torch.empty((n, c, t, h, w), dtype=dtype, device=gpu)
    a_sums = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    dw = torch.full((c, 27), 0.25, dtype=torch.float32, device=gpu)
    ops.dw3d_bwd(dv, braw, coef, araw, ss, wt.view(c, 27), ga, a_sums, dw, 2)
    torch.cuda.synchronize()
    name = hip.kernel_name(S.dw_bwd_struct((n, c, t, h, w, 2), dtype))
    assert "s2r" in name, name
    wd = wt.double()
    dw_ref = torch.zeros((c, 3, 3, 3), dtype=torch.float64, device=gpu)
    worst, CH = 0.0, 4
    s1 = torch.zeros(c, dtype=torch.float64, device=gpu)
    s2 = torch.zeros(c, dtype=torch.float64, device=gpu)
    rt, at = tol_store(dtype)
    for i in range(0, n, CH):
        cd = coef[i:i + CH].double()
        dB = cd[:, :, 0, None, None, None] * dv[i:i + CH].double() + cd[:, :, 1, None, None, None] * braw[i:i + CH].double() + cd[:, :, 2, None, None, None]
        z = araw[i:i + CH].double() * ss[:, 0].double().view(1, -1, 1, 1, 1) + ss[:, 1].double().view(1, -1, 1, 1, 1)
        pa = torch.nn.functional.pad(torch.relu(z), (0, 1, 0, 1, 1, 1))
        dAp = torch.zeros_like(pa)
        for kt in range(3):
            for kh in range(3):
                for kw in range(3):
                    # out[t, ho, wo] = sum_k w[k] a[t + kt - 1, 2 ho + kh, 2 wo + kw]
                    sl = (slice(None), slice(None), slice(kt, kt + t), slice(kh, kh + 2 * ho, 2), slice(kw, kw + 2 * wo, 2))
                    dAp[sl] += wd[:, kt, kh, kw].view(1, -1, 1, 1, 1) * dB
                    dw_ref[:, kt, kh, kw] += (dB * pa[sl]).sum((0, 2, 3, 4))
        ref = dAp[:, :, 1:1 + t, :h, :w] * (z > 0)
        worst = max(worst, _check(f"ga[{i}:{i + CH}]", ga[i:i + CH], ref, rt, at * ref.abs().max().item()))
        gs = ga[i:i + CH].double()
        s1 += gs.sum((0, 2, 3, 4))
        s2 += (gs * araw[i:i + CH].double()).sum((0, 2, 3, 4))
        del dB, z, pa, dAp, ref, gs
    e = _check("dw", dw.double().view(c, 3, 3, 3) - 0.25, dw_ref, 2e-5, 2e-5 * dw_ref.abs().max().item())
    st = 3e-3 if dtype == torch.bfloat16 else 5e-4
    sref = torch.stack([s1, s2], 1)
    _check("a_sums", a_sums, sref, 10 * st, 10 * st * max(1.0, sref.abs().max().item()))
    print(f"full-size 112^2 -> 56^2 depthwise backward {dtype} ({name}): ga err {worst:.2e} of max, dW err {e:.2e} of max (limit 2e-05)")
