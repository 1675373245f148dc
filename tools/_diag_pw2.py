import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (cin, cout, hw) in [(216, 96, 14), (108, 48, 28), (54, 24, 56), (96, 216, 14)]:
    n, t = 64, 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn((n, cin, t, hw, hw), generator=g).bfloat16().to(dev)
    w = (torch.randn((cout, cin), generator=g) * 0.1).to(dev)
    (fp, dp), = ops.pw_pack_weights([w])
    ss = torch.stack([1 + 0.3 * torch.randn(cin, generator=g), 0.3 * torch.randn(cin, generator=g)], 1).to(dev)
    gate = torch.rand((n, cin), generator=g).to(dev)
    y = torch.empty((n, cout, t, hw, hw), dtype=torch.bfloat16, device=dev)
    res = {}
    res["none"] = timed(lambda: ops.pw_fwd(x, w, y=y, w_panel=fp))
    res["affine"] = timed(lambda: ops.pw_fwd(x, w, y=y, in_ss=ss, in_act=0, w_panel=fp))
    res["relu"] = timed(lambda: ops.pw_fwd(x, w, y=y, in_ss=ss, in_act=1, w_panel=fp))
    res["swish"] = timed(lambda: ops.pw_fwd(x, w, y=y, in_ss=ss, in_act=2, w_panel=fp))
    res["swish+gate"] = timed(lambda: ops.pw_fwd(x, w, y=y, in_ss=ss, in_gate=gate, in_act=2, w_panel=fp))
    print(f"{cin}->{cout} @{hw}x{hw}: " + "  ".join(f"{k} {v:.1f}" for k, v in res.items()))
