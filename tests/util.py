import torch


def report(name, got, ref, rtol, atol):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        idx = torch.nonzero(bad | ~torch.isfinite(got))[0].tolist() if (bad | ~torch.isfinite(got)).any() else None
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} elements out of tolerance (rtol={rtol}, atol={atol}); "
            f"max abs err {err.max().item():.3e}, max |ref| {ref.abs().max().item():.3e}; first bad index {idx}: "
            f"got {got[tuple(idx)].item() if idx is not None else None} ref {ref[tuple(idx)].item() if idx is not None else None}")
    return err.max().item()


def tie_slack(v, dtype, w_abs, eps_ulps=2e-4):
    """Extra absolute tolerance [N, M, T, H, W] for a GEMM whose operand `v` ([N, K, T, H, W], the fp32 output of a prologue)
    is rounded to the 16-bit `dtype` before the product.  Where a value sits within `eps_ulps` of the midpoint between two
    neighbouring `dtype` values, fp32 arithmetic in another order (the kernel's folded swish against torch's) may round it to
    the other neighbour: one ulp of the operand, so every output at that point may move by ulp * |w[m, k]|.  Zero almost
    everywhere.  (Found by the fuzz sweep, seed 20261004: a prologue value 2.1e-6 bf16-ulps off a midpoint under a weight
    of 0.66 moved 15 outputs of one point by up to 1.2e-2, five times the flat atol.)"""
    v64 = v.detach().double().cpu()
    r = v64.float().to(dtype).double()
    mant = 7 if dtype == torch.bfloat16 else 10
    ulp = torch.exp2(torch.floor(torch.log2(r.abs().clamp_min(1e-30))) - mant)
    d = (v64 - r).abs() / ulp                         # 0 .. 0.5 (0.5 = a tie)
    amb = ((0.5 - d) < eps_ulps).double() * ulp       # [N, K, T, H, W]
    return torch.einsum("mk,nkthw->nmthw", w_abs.detach().double().cpu(), amb)


def relu_mask_mismatch(masks, oracle_taps_masks):
    """Fraction of ReLU sites whose sign differs between the device (hip_relu_masks) and a free-running oracle forward."""
    bad = tot = 0
    for k, m in oracle_taps_masks.items():
        d = masks[k]
        bad += int((d != m).sum())
        tot += m.numel()
    return bad / max(tot, 1), bad, tot


def hip_relu_masks(pl):
    """Sign patterns of every ReLU in a training plan of the HIP model, keyed like the oracle's ReLU sites.
    Computed in fp64 from the stored fp32 tensors/coefficients: the kernels evaluate s*x+t with one fused
    rounding, which preserves the sign of the exact value."""
    def affine_mask(raw, ss):
        ss = ss.detach().double().cpu()
        z = raw.detach().double().cpu() * ss[:, 0].view(1, -1, 1, 1, 1) + ss[:, 1].view(1, -1, 1, 1, 1)
        return z > 0

    masks = {"conv1": pl.y0.detach().float().cpu() > 0,
             "conv5": affine_mask(pl.c5_raw, pl.bn5.ss),
             "fc1": pl.h1.detach().cpu() > 0}
    for B in pl.blocks:
        s = B.spec
        pre = f"stages/{s.stage}/stage/layer_with_weights-{s.index}"
        masks[pre + "/a"] = affine_mask(B.a_raw, B.bn_a.ss)
        masks[pre + "/out"] = B.y.detach().float().cpu() > 0
        if s.has_se:
            masks[pre + "/se"] = B.hidden.detach().cpu() > 0
    return masks


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-30)).item()


def round_to(x, dtype):
    """x (fp32 / fp64) rounded to the storage type `dtype`, as fp64: what a 16-bit GEMM operand holds after the kernel's
    fp32 prologue.  float32: unchanged (the fp32 kernels feed the exact-fp32 matrix instruction)."""
    if dtype == torch.float32:
        return x.double()
    return x.float().to(dtype).double()


def tol_gemm(dtype):
    """(rtol, atol-per-unit-of-scale) for the pointwise (matrix-core) kernels against an fp64 GEMM whose OPERANDS were
    rounded to the storage type first (weights; the fp32 prologue's output -- round_to), so that what is left is the
    rounding of the stored output (rtol, as tol_store), fp32 accumulation order, and the rare operand whose fp32
    prologue value sits on a rounding boundary and lands on the other side in the reference's arithmetic (one product
    off by an operand ulp: << 1e-3 of the tensor's scale).  Round 2 compared with UNROUNDED operands at 1.6e-2 of the
    tensor maximum, which a dropped k-element of a K = 432 GEMM (1.2 %) passed."""
    if dtype == torch.float32:
        return (2e-5, 2e-5)
    return (4e-3, 1e-3) if dtype == torch.bfloat16 else (1e-3, 2.5e-4)


def tol_store(dtype):
    """(rtol, atol-per-unit-of-scale) for kernels whose ONLY error source against the fp64 reference is the rounding of
    the stored output (depthwise convs, residual tails, stem temporal conv: inputs are pre-rounded, arithmetic fp32):
    half an ulp of the storage type relative (2^-9 bf16, 2^-12 fp16... 2^-11 as fp16 has 11 significand bits) plus fp32
    accumulation noise."""
    if dtype == torch.float32:
        return (2e-5, 2e-5)
    return (4e-3, 2e-4) if dtype == torch.bfloat16 else (1e-3, 1e-4)


def rnd(shape, dtype, gen, scale=1.0):
    """random tensor representable in `dtype`, returned as (device-dtype tensor on cpu, fp64 copy)"""
    t = (torch.randn(shape, generator=gen, dtype=torch.float32) * scale).to(dtype)
    return t, t.double()


class AltBackward:
    """A second backward launch list recorded over the SAME forward buffers of a training plan with other plan options
    (record_alternate_backward).  run() replays it from the forward state `snapshot` captured."""

    def __init__(self, model, pl, lst, extra):
        self.model, self.pl, self.lst, self.extra = model, pl, lst, extra

    def run(self):
        self.extra.zero_()
        self.pl.run(self.lst)


def record_alternate_backward(model, pl, x, **options):
    """Differential tests: record the backward pass of `pl` AGAIN with `options` overriding the model's plan options, against
    the forward tensors the plan already owns -- so that two backward variants (e.g. pw_bwd_rc on / off) can be compared on
    bit-identical forward state, where the backward pass is a LINEAR map of the upstream gradient and differences do not
    amplify.  New fp64 accumulators of the second list live in their own zeroed buffer; x = the bound input batch (the
    stem's weight-gradient launch of the new list is bound to it).  Test-side only: pokes at plan internals."""
    import torch
    from x3d_tf_amd.model import PLAN_DEFAULTS, X3D
    assert all(k in PLAN_DEFAULTS for k in options)
    saved_opt, saved_fuse, saved_rc = model.opt, model._fuse_pw_bwd, model._rc_pw_bwd
    saved_bwd, saved_marks = pl.bwd, dict(pl.bwd_stage_marks)
    saved_pl = {k: getattr(pl, k, None) for k in ("gbuf", "dv", "ga", "rtmp", "coef_nc", "se_scratch", "g5", "dh1", "dpooled",
                                                  "ds", "stem_bwd_folded")}
    saved_b = [{k: getattr(B, k, None) for k in ("bwd_start", "bwd_stop", "dy_view", "dx_view", "tail_folded", "a_bwd_rc",
                                                 "r_bwd_rc", "db", "nc_sums")} for B in pl.blocks]
    try:
        model.opt = dict(model.opt, **{k: bool(v) for k, v in options.items()})
        model._fuse_pw_bwd, model._rc_pw_bwd = model.opt["fused_pw_bwd"], model.opt["pw_bwd_rc"]
        for B in pl.blocks:
            B.tail_folded = False
        pl.bwd, pl.bwd_stage_marks = [], {}
        n0 = len(pl._zero_chunks)
        model._record_backward(pl)
        alt = pl.bwd
        new = pl._zero_chunks[n0:]
        extra = torch.zeros(max(sum(c[0] for c in new), 1), dtype=torch.float64, device=model.device)
        off = 0
        for numel, shape in new:
            pl._zero_views.append(extra[off:off + numel].view(shape))
            off += numel
        X3D._resolve(pl, alt)
        for f, handle in getattr(pl, "bwd_folds", []):          # (folds recorded by either list: the same accumulators)
            f.sums = pl._zero_views[handle].data_ptr()
        info = dict(tail_folded=[bool(B.tail_folded) for B in pl.blocks], a_bwd_rc=[bool(getattr(B, "a_bwd_rc", False)) for B in pl.blocks],
                    stem_bwd_folded=bool(getattr(pl, "stem_bwd_folded", False)))
        # the scratch buffers of the new list are referenced from argument structs by ADDRESS only (the plan keeps them alive as
        # attributes, which are restored to the first list's below): hold them here, or the allocator hands their memory out again
        owned = [getattr(pl, k, None) for k in saved_pl] + [extra]
    finally:
        model.opt, model._fuse_pw_bwd, model._rc_pw_bwd = saved_opt, saved_fuse, saved_rc
        pl.bwd, pl.bwd_stage_marks = saved_bwd, saved_marks
        for k, v in saved_pl.items():
            setattr(pl, k, v)
        for B, d in zip(pl.blocks, saved_b):
            for k, v in d.items():
                setattr(B, k, v)
    model._bind_input(pl, x)          # (the input slots include the new list's stem launch)
    ab = AltBackward(model, pl, alt, extra)
    ab.info, ab.owned = info, owned
    return ab
