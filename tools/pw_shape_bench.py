"""Times x3d_pw_fwd on single layer shapes (packed panels, 16-bit storage): the A/B harness for dispatch experiments
(X3D_PW_MTMAP, X3D_PW_TPBMIN, X3D_PW_WS ... are read once per process, so each setting is its own run).

    python tools/pw_shape_bench.py [fp16|bf16] N,Cin,Cout,T,H,W,pro,res ...     pro: n|s (swish prologue)  res: n|i|c
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import hip, ops  # noqa: E402


def main():
    dtype = torch.float16 if sys.argv[1] == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for spec in sys.argv[2:]:
        f = spec.split(",")
        n, cin, cout, t, h, w = map(int, f[:6])
        pro, res = f[6], f[7]
        x = torch.randn((n, cin, t, h, w), generator=g, device=dev).to(dtype)
        wt = torch.randn((cout, cin), generator=g, device=dev) * 0.1
        (fp, dp), = ops.pw_pack_weights([wt], dtype=dtype)
        kw = {}
        if pro == "s":
            kw.update(in_ss=torch.rand((cin, 2), generator=g, device=dev), in_gate=torch.rand((n, cin), generator=g, device=dev), in_act=2)
        if res != "n":
            kw.update(out_ss=torch.rand((cout, 2), generator=g, device=dev), out_act=1,
                      out_add=torch.randn((n, cout, t, h, w), generator=g, device=dev).to(dtype))
            if res == "c":
                kw.update(out_add_ss=torch.rand((cout, 2), generator=g, device=dev))
        else:
            kw.update(stats=None)
        y = torch.empty((n, cout, t, h, w), dtype=dtype, device=dev)
        for _ in range(3):
            ops.pw_fwd(x, wt, y=y, w_panel=fp, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.pw_fwd(x, wt, y=y, w_panel=fp, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        by = 2 * n * t * h * w * (cin + cout + (cout if res != "n" else 0))
        fl = 2.0 * n * t * h * w * cin * cout
        st = hip.PwFwdArgs(x.data_ptr(), wt.data_ptr(), y.data_ptr(), None, kw["in_ss"].data_ptr() if pro == "s" else None,
                           kw["in_gate"].data_ptr() if pro == "s" else None, 2 if pro == "s" else 0, n, cin, cout, t, h, w, 1,
                           hip.dtype_code(dtype), fp.data_ptr(), out_scale_shift=kw["out_ss"].data_ptr() if res != "n" else None,
                           out_add=kw["out_add"].data_ptr() if res != "n" else None,
                           out_add_scale_shift=kw["out_add_ss"].data_ptr() if res == "c" else None, out_act=1 if res != "n" else 0)
        print(f"{spec:34s} {us:8.1f} us  {by / us / 1e6:6.2f} TB/s  {fl / us / 1e6:7.1f} TFLOP/s  {hip.pw_kernel_name(st)}", flush=True)


if __name__ == "__main__":
    main()
