"""Times the stem both ways on one shape: the four launches of the two-kernel path (x3d_stem_s_fwd, x3d_dwt_fwd, x3d_dwt_bwd,
x3d_stem_s_wgrad) against the fused pair (x3d_stem_fwd, x3d_stem_bwd), with the bytes each form moves.

    python tools/stem_bench.py [bf16|fp16] [N T H W C1]          default: bf16 64 16 224 224 24 (the headline's stem)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import ops  # noqa: E402


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    a = sys.argv[1:]
    dtype = torch.float16 if a and a[0] == "fp16" else torch.bfloat16
    if a and a[0] in ("fp16", "bf16"):
        a = a[1:]
    n, t, h, w, c1 = (int(v) for v in a) if a else (64, 16, 224, 224, 24)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((n, t, h, w, 3), generator=g, device=dev).to(dtype)
    ws = torch.randn((c1, 3, 3, 3), generator=g, device=dev) * 0.3
    wt = torch.randn((c1, 5), generator=g, device=dev) * 0.4
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ys = torch.empty((n, c1, t, ho, wo), dtype=dtype, device=dev)
    yt = torch.empty_like(ys)
    ds = torch.empty_like(ys)
    gr = torch.randn(ys.shape, generator=g, device=dev).to(dtype)
    coef = torch.randn((c1, 4), generator=g, device=dev) * 0.5
    rss = torch.stack([1 + 0.3 * torch.randn(c1, generator=g, device=dev), 0.3 * torch.randn(c1, generator=g, device=dev)], 1).contiguous()
    st = ops.stats_buffer(c1, dev)
    dwt = torch.zeros((c1, 5), device=dev)
    dws = torch.zeros((c1, 3, 3, 3), device=dev)
    xb, yb = x.numel() * 2, ys.numel() * 2
    rows = [
        ("x3d_stem_s_fwd", lambda: ops.stem_s_fwd(x, ws, y=ys, channels_last=True), xb + yb),
        ("x3d_dwt_fwd", lambda: ops.dwt_fwd(ys, wt, y=yt, stats=st), 2 * yb),
        ("x3d_dwt_bwd", lambda: ops.dwt_bwd(gr, yt, coef, ys, wt, ds, dwt, relu_ss=rss), 4 * yb),
        ("x3d_stem_s_wgrad", lambda: ops.stem_s_wgrad(x, ds, dws, channels_last=True), xb + yb),
        ("x3d_stem_fwd", lambda: ops.stem_fwd(x, ws, wt, y=yt, stats=st), xb + yb),
        ("x3d_stem_bwd", lambda: ops.stem_bwd(gr, yt, coef, x, ws, wt, dws, dwt, relu_ss=rss), xb + 2 * yb),
    ]
    tot = {}
    for name, fn, by in rows:
        us = timed(fn)
        tot[name] = us
        print(f"{name:18s} {us:8.1f} us  {by / 1e6:8.1f} MB  {by / us / 1e6:5.2f} TB/s", flush=True)
    two = sum(tot[k] for k in ("x3d_stem_s_fwd", "x3d_dwt_fwd", "x3d_dwt_bwd", "x3d_stem_s_wgrad"))
    one = tot["x3d_stem_fwd"] + tot["x3d_stem_bwd"]
    print(f"two-kernel path {two:.1f} us, fused {one:.1f} us  ({n}x{t}x{h}x{w}, {c1} channels, {str(dtype).split('.')[-1]})")


if __name__ == "__main__":
    main()
