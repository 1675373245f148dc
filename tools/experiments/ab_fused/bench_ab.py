"""The `a` conv without its output tensor against the stored-a_raw pair, layer by layer at X3D-M's sizes (64 clips, bf16):
x3d_pw_gram + x3d_ab_fwd   vs   x3d_pw_fwd (+ statistics) + x3d_dw3d_fwd.     python tools/bench_ab.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
LAYERS = [("s2 b0", 24, 54, 16, 112, 112, 2), ("s2 b1", 24, 54, 16, 56, 56, 1), ("s3 b0", 24, 108, 16, 56, 56, 2),
          ("s3 b1", 48, 108, 16, 28, 28, 1), ("s4 b0", 48, 216, 16, 28, 28, 2)]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for name, cin, c, t, h, w, s in LAYERS:
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((B, cin, t, h, w), generator=g, device=dev).relu().bfloat16()
    wa = (torch.randn((c, cin), generator=g, device=dev) * 0.2)
    wb = (torch.randn((c, 27), generator=g, device=dev) * 0.2)
    ss = torch.stack([torch.ones(c, device=dev), torch.zeros(c, device=dev)], 1).contiguous()
    (fp, dp), = ops.pw_pack_weights([wa], dtype=torch.bfloat16)
    ho, wo = -(-h // s), -(-w // s)
    a_raw = torch.empty((B, c, t, h, w), dtype=torch.bfloat16, device=dev)
    y = torch.empty((B, c, t, ho, wo), dtype=torch.bfloat16, device=dev)
    st_a = ops.stats_buffer(c, dev)
    st_b = ops.stats_buffer(c, dev)
    pool = torch.zeros((B, c), dtype=torch.float64, device=dev)
    gram = None
    t_pw = timeit(lambda: ops.pw_fwd(x, wa, y=a_raw, stats=st_a, w_panel=fp))
    t_dw = timeit(lambda: ops.dw3d_fwd(a_raw, wb, s, y=y, in_ss=ss, in_act=1, stats=st_b, pool=pool))
    gram = ops.pw_gram(x)
    t_gr = timeit(lambda: ops.pw_gram(x, gram))
    y2 = torch.empty_like(y)
    ok = ops.ab_fwd(x, wa, ss, wb, s, y=y2, stats=st_b, pool=pool)
    t_ab = timeit(lambda: ops.ab_fwd(x, wa, ss, wb, s, y=y2, stats=st_b, pool=pool)) if ok is not None else float("nan")
    err = (y.float() - y2.float()).abs().max().item() if ok is not None else float("nan")
    xb, ab, yb = x.numel() * 2 / 1e6, a_raw.numel() * 2 / 1e6, y.numel() * 2 / 1e6
    print(f"{name} {cin}->{c} @{t}x{h}x{w} s{s}: pw_fwd {t_pw:7.1f} us + dw3d_fwd {t_dw:7.1f} us = {t_pw + t_dw:7.1f} | "
          f"gram {t_gr:6.1f} us ({xb / t_gr * 1e3:5.0f} GB/s) + ab_fwd {t_ab:7.1f} us ({(xb + yb) / t_ab * 1e3:5.0f} GB/s) = {t_gr + t_ab:7.1f} | "
          f"max |y - y_fused| {err:.3g} (a_raw rounded to bf16 on the left only)", flush=True)
