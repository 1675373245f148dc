"""The two parity tests of the experiment as they stood in tests/test_kernels_gpu.py (ABI 127)."""
# --------------------------------------------------------------------------------------------------
# the `a` conv without its output tensor (ab_fused.hip)
@pytest.mark.parametrize("dtype", S.HALF_DTYPES)
@pytest.mark.parametrize("shape", [(2, 24, 3, 16, 16, None), (3, 24, 2, 28, 28, "stem"), (2, 24, 4, 8, 8, "tail"),
                                   (2, 48, 1, 12, 12, "tail_conv"), (1, 32, 2, 16, 8, None), (1, 8, 1, 40, 40, "tail"),
                                   (1, 64, 1, 8, 8, None), (5, 24, 8, 28, 28, "tail_conv")])
def test_pw_gram(gpu, dtype, shape):
    """x3d_pw_gram: [x ; 1] x^T over all points against fp64, with each prologue that builds x on load (stored x bit for bit
    = the rounded fp32 expression), and x3d_bn_finalize_gram: the BatchNorm coefficients of a = W x they imply against the
    fp64 statistics of the explicit product."""
    ops = _ops()
    n, cin, t, h, w, pro = shape
    g_ = _gen(41)
    if pro is None:
        x, xd = rnd((n, cin, t, h, w), dtype, g_)
        gram = ops.pw_gram(x.to(gpu))
    else:
        raw, rawd = rnd((n, cin, t, h, w), dtype, g_)
        ss1 = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1)
        add = addd = ss2 = None
        v = _affine(rawd, ss1.double())
        if pro != "stem":
            add, addd = rnd((n, cin, t, h, w), dtype, g_)
            if pro == "tail_conv":
                ss2 = torch.stack([1 + 0.3 * torch.randn(cin, generator=g_), 0.3 * torch.randn(cin, generator=g_)], 1)
                v = v + _affine(addd, ss2.double())
            else:
                v = v + addd
        xref = F.relu(v)
        xg = torch.empty((n, cin, t, h, w), dtype=dtype, device=gpu)
        dev = lambda q: None if q is None else q.to(gpu)
        gram = ops.pw_gram(xg, raw=dev(raw), raw_ss=dev(ss1), add=dev(add), add_ss=dev(ss2))
        rt, at = tol_store(dtype)
        report("built x", xg, xref, rt, at * xref.abs().max().item())
        xd = xg.float().cpu().double()              # the Gram matrix describes x as stored
    torch.cuda.synchronize()
    ref = torch.cat([torch.einsum("ncthw,ndthw->cd", xd, xd), xd.sum((0, 2, 3, 4))[None]], 0)
    report("gram", gram.sum(0), ref, 1e-5, 1e-5 * ref.abs().max().item())       # (the copies of the replicated layout added up)
    # BN coefficients of a = W x
    c = 2 * cin + 6
    wt = torch.randn((c, cin), generator=g_) * 0.3
    gamma = 1 + 0.2 * torch.randn(c, generator=g_)
    beta = 0.2 * torch.randn(c, generator=g_)
    ss = torch.empty((c, 2), device=gpu)
    mi = torch.empty((c, 2), device=gpu)
    mm, mv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
    M = n * t * h * w
    ops.bn_finalize_gram(gram, wt.to(gpu), M, gamma.to(gpu), beta.to(gpu), mm, mv, 1e-5, 0.9, True, ss, mi, dtype)
    torch.cuda.synchronize()
    ad = torch.einsum("oc,ncthw->nothw", round_to(wt, dtype), xd)
    mean, var = ad.mean((0, 2, 3, 4)), ad.var((0, 2, 3, 4), unbiased=False)
    inv = 1 / torch.sqrt(var + 1e-5)
    report("scale", ss[:, 0], gamma.double() * inv, 2e-5, 1e-6)
    report("shift", ss[:, 1], beta.double() - mean * gamma.double() * inv, 2e-5, 2e-5)
    report("mean", mi[:, 0], mean, 2e-5, 1e-5)
    report("moving_var", mv, 0.9 + 0.1 * var * M / (M - 1), 2e-5, 1e-6)


@pytest.mark.parametrize("dtype", S.HALF_DTYPES)
@pytest.mark.parametrize("shape", S.AB)
def test_ab_fwd(gpu, dtype, shape):
    """x3d_ab_fwd (a -> bn_a -> relu -> b fused, the `a` output never stored) against the fp64 composition of the three ops
    on the same stored input and weights rounded as the matrix cores see them.  a stays in fp32 between the product and the
    stencil, so what is left is the rounding of the stored output (tol_store) and fp32 summation order."""
    ops = _ops()
    n, cin, c, t, h, w, stride = shape
    g_ = _gen(43)
    x, xd = rnd((n, cin, t, h, w), dtype, g_)
    wa = torch.randn((c, cin), generator=g_) * 0.3
    wb = torch.randn((c, 3, 3, 3), generator=g_) * 0.3
    ss = torch.stack([1 + 0.3 * torch.randn(c, generator=g_), 0.3 * torch.randn(c, generator=g_)], 1)
    ho, wo = -(-h // stride), -(-w // stride)
    stats = torch.zeros((c, 2), dtype=torch.float64, device=gpu)
    pool = torch.zeros((n, c), dtype=torch.float64, device=gpu)
    y = ops.ab_fwd(x.to(gpu), wa.to(gpu), ss.to(gpu), wb.to(gpu), stride, stats=stats, pool=pool)
    torch.cuda.synchronize()
    assert y is not None, "the fused forward should cover this shape"
    from oracle import x3d_oracle as O
    ad = torch.einsum("oc,ncthw->nothw", round_to(wa, dtype), xd)
    act = F.relu(_affine(ad, ss.double()))
    ref = O.depthwise3x3x3(act, wb.double().view(c, 27), stride)
    assert tuple(y.shape) == (n, c, t, ho, wo)
    rt, at = tol_store(dtype)
    report("y", y, ref, rt, 2 * at * ref.abs().max().item())
    ys = y.float().cpu()                     # the statistics describe the tensor as stored (as x3d_dw3d_fwd's do)
    sref = _stats_ref(ys, dtype)
    report("stats", stats, sref, _stol(dtype), _stol(dtype) * max(1.0, sref.abs().max().item()))
    report("pool", pool, ys.double().sum((2, 3, 4)), _stol(dtype), 10 * _stol(dtype) * max(1.0, float(ys[0, 0].numel()) ** 0.5))


