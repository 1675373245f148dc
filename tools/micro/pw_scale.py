"""Time x3d_pw_fwd / x3d_pw_dgrad on a stage-5 shape against the batch size: the slope is the per-tile cost of the
persistent kernel, the intercept its fixed cost.   python tools/micro/pw_scale.py [cin cout]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import x3d_tf_amd as x  # noqa: E402
from x3d_tf_amd import ops  # noqa: E402

cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (432, 192)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
wt = (torch.randn((cout, cin), generator=g) * 0.1).to(dev)
(fp, dp), = ops.pw_pack_weights([wt])
ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g)], 1).to(dev)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


for n in (16, 32, 64, 128, 256):
    xin = torch.randn((n, cin, 16, 7, 7), generator=g).bfloat16().to(dev)
    gate = torch.rand((n, cin), generator=g).to(dev)
    st = ops.stats_buffer(cout, dev) if hasattr(ops, "stats_buffer") else torch.zeros((cout, 2), dtype=torch.float64, device=dev)
    y = torch.empty((n, cout, 16, 7, 7), dtype=torch.bfloat16, device=dev)
    t_none = timeit(lambda: ops.pw_fwd(xin, wt, stats=st, w_panel=fp, y=y))
    t_sw = timeit(lambda: ops.pw_fwd(xin, wt, stats=st, w_panel=fp, in_ss=ss, in_gate=gate, in_act=2, y=y))
    t_relu = timeit(lambda: ops.pw_fwd(xin, wt, stats=st, w_panel=fp, in_ss=ss, in_act=1, y=y))
    print(f"N={n:4d} tiles={25 * n:5d}  fwd none {t_none:7.1f} us   relu {t_relu:7.1f} us   swish+gate {t_sw:7.1f} us", flush=True)
