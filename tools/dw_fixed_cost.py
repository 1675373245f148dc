"""How much of a ragged-plane depthwise forward launch is per-workgroup fixed cost?  The same bytes as (N, C, T) = (60, 162, 16),
(30, 162, 32), (120, 162, 8) on 39x39 planes (X3D-XL stage 3, inference form: BN + ReLU prologue, pool sums, no statistics) and the
neighbouring even sizes 40x40 / 36x36, fp16.   python tools/dw_fixed_cost.py        (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [(60, 162, 16, 39, 39), (30, 162, 32, 39, 39), (120, 162, 8, 39, 39), (60, 162, 16, 40, 40), (60, 162, 16, 36, 36), (60, 162, 16, 39, 40), (60, 162, 16, 40, 39),
                        (60, 72, 16, 78, 78), (60, 72, 16, 80, 80), (60, 306, 16, 20, 20)]
if len(sys.argv) > 1:      # python tools/dw_fixed_cost.py 38x39 41x39 ...   (H x W, N = 60, C = 162, T = 16)
    SHAPES = [(60, 162, 16, int(a.split('x')[0]), int(a.split('x')[1])) for a in sys.argv[1:]]
for (n, c, t, h, w) in SHAPES:
    x = torch.randn((n, c, t, h, w), generator=g, device=dev).half()
    wt = torch.randn((c, 3, 3, 3), generator=g, device=dev) * 0.2
    ss = torch.randn((c, 2), generator=g, device=dev)
    y = torch.empty_like(x)
    pool = torch.zeros((n, c), dtype=torch.float64, device=dev)
    for name, fn in (("prologue+pool", lambda: ops.dw3d_fwd(x, wt, 1, y=y, in_ss=ss, in_act=1, pool=pool)), ("bare", lambda: ops.dw3d_fwd(x, wt, 1, y=y))):
        us = timed(fn)
        by = 2 * x.numel() * 2
        print(f"N={n:3d} C={c} T={t:2d} {h}x{w}  {name:14s} {us:8.1f} us  {by / us / 1e6:7.1f} GB/s", flush=True)
