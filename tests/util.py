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


def tol_for(dtype):
    # (rtol, atol) for outputs stored in `dtype`; inputs are pre-rounded so only output rounding and
    # fp32 accumulation order differ from the fp64 reference
    return (2e-5, 2e-5) if dtype == torch.float32 else (1.6e-2, 1.6e-2)


def rnd(shape, dtype, gen, scale=1.0):
    """random tensor representable in `dtype`, returned as (device-dtype tensor on cpu, fp64 copy)"""
    t = (torch.randn(shape, generator=gen, dtype=torch.float32) * scale).to(dtype)
    return t, t.double()
