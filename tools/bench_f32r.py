"""fp32 pointwise layers of BASELINE config 2 in isolation (x3d_pw_fwd / x3d_pw_dgrad, fp32 storage).  python tools/bench_f32r.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import ops
dev = torch.device("cuda:0")
N = 32
def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
g = torch.Generator(device=dev).manual_seed(0)
for name, cin, cout, t, h, w, kind in [("a bwd s5", 192, 432, 13, 5, 5, "dgrad_add"), ("c bwd s5", 432, 192, 13, 5, 5, "dgrad_swish"),
                                        ("a bwd s4", 96, 216, 13, 10, 10, "dgrad_add"), ("c bwd s4", 216, 96, 13, 10, 10, "dgrad_swish"),
                                        ("c fwd s5", 432, 192, 13, 5, 5, "fwd_swish"), ("a fwd s5", 192, 432, 13, 5, 5, "fwd"),
                                        ("c fwd s4", 216, 96, 13, 10, 10, "fwd_swish"), ("a fwd s4", 96, 216, 13, 10, 10, "fwd"),
                                        ("c fwd s3", 108, 48, 13, 20, 20, "fwd_swish"), ("a fwd s3", 48, 108, 13, 20, 20, "fwd"),
                                        ("c bwd s3", 108, 48, 13, 20, 20, "dgrad_swish"), ("a bwd s3", 48, 108, 13, 20, 20, "dgrad_add"),
                                        ("c bwd s2", 54, 24, 13, 40, 40, "dgrad_swish"), ("a bwd s2", 24, 54, 13, 40, 40, "dgrad_add")]:
    wt = torch.randn((cout, cin), generator=g, device=dev) * 0.1
    if kind.startswith("dgrad"):
        gy = torch.randn((N, cout, t, h, w), generator=g, device=dev)
        yraw = torch.randn((N, cout, t, h, w), generator=g, device=dev)
        coef = torch.randn((cout, 4), generator=g, device=dev)
        dx = torch.empty((N, cin, t, h, w), device=dev)
        if kind == "dgrad_add":
            add = torch.randn((N, cin, t, h, w), generator=g, device=dev)
            fn = lambda: ops.pw_dgrad(gy, yraw, coef, wt, dx, ops.EPI_ADD, add=add)
        else:
            braw = torch.randn((N, cin, t, h, w), generator=g, device=dev)
            bss = torch.randn((cin, 2), generator=g, device=dev)
            gate = torch.rand((N, cin), generator=g, device=dev)
            ncs = torch.zeros((N, cin, 2), dtype=torch.float64, device=dev)
            fn = lambda: ops.pw_dgrad(gy, yraw, coef, wt, dx, ops.EPI_SWISH_BWD, braw=braw, b_ss=bss, gate=gate, nc_sums=ncs)
        flops = 2.0 * cin * cout * N * t * h * w
    else:
        x = torch.randn((N, cin, t, h, w), generator=g, device=dev)
        y = torch.empty((N, cout, t, h, w), device=dev)
        st = ops.stats_buffer(cout, dev)
        if kind == "fwd_swish":
            ss = torch.randn((cin, 2), generator=g, device=dev)
            gate = torch.rand((N, cin), generator=g, device=dev)
            fn = lambda: ops.pw_fwd(x, wt, y=y, stats=st, in_ss=ss, in_gate=gate, in_act=2)
        else:
            fn = lambda: ops.pw_fwd(x, wt, y=y, stats=st)
        flops = 2.0 * cin * cout * N * t * h * w
    us = timeit(fn)
    print(f"{name:9s} {kind:12s} {cin:4d}->{cout:4d} @{t}x{h}x{w}: {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s", flush=True)
