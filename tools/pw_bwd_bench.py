"""Times the pointwise BACKWARD launches of one layer shape in isolation, with environment variants interleaved in ONE
process (experiments build: the library re-reads X3D_* switches on every call).

    X3D_EXPERIMENTS=1 python x3d-tf_amd/build.py && \
    python tools/pw_bwd_bench.py bf16 c,64,216,96,16,14,14 a,64,96,216,16,14,14 -- "" "X3D_PW_BWD_NOFLUSH=1"

layer = kind,N,Cin,Cout,T,H,W   kind: c = `c` conv (swish' epilogue, per-(n,c) sums), a = `a` conv (identity-shortcut add),
        at = `a` conv with the folded tail; the fused x3d_pw_bwd where it covers the shape, else x3d_pw_wgrad + x3d_pw_dgrad
        (both timed, sum reported).  Variants after `--`: space-separated NAME=VALUE lists ("" = defaults).
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import hip, ops  # noqa: E402


def build(kind, n, cin, cout, t, h, w, dtype, dev, g):
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    gy = rn(n, cout, t, h, w).to(dtype)
    yraw = rn(n, cout, t, h, w).to(dtype)
    coef = rn(cout, 4) * 0.5
    wt = rn(cout, cin) * 0.1
    (fp, dp), = ops.pw_pack_weights([wt], dtype=dtype)
    dx = torch.empty((n, cin, t, h, w), dtype=dtype, device=dev)
    dw = torch.zeros((cout, cin), dtype=torch.float32, device=dev)
    kw = {}
    if kind == "c":
        x = rn(n, cin, t, h, w).to(dtype)                     # b_raw
        ss = torch.rand((cin, 2), generator=g, device=dev)
        gate = torch.rand((n, cin), generator=g, device=dev)
        ncs = torch.zeros((n, cin, 2), dtype=torch.float64, device=dev)
        fused = lambda: ops.pw_bwd(gy, yraw, coef, dp, dx, dw, ops.EPI_SWISH_BWD, braw=x, b_ss=ss, gate=gate, nc_sums=ncs)
        wg = lambda: ops.pw_wgrad(gy, yraw, coef, x, dw, in_ss=ss, in_gate=gate, in_act=2)
        dg = lambda: ops.pw_dgrad(gy, yraw, coef, wt, dx, ops.EPI_SWISH_BWD, braw=x, b_ss=ss, gate=gate, nc_sums=ncs, w_panel=dp)
        by = 2 * n * t * h * w * (2 * cout + 2 * cin)
    else:
        x = torch.relu(rn(n, cin, t, h, w)).to(dtype)
        add = rn(n, cin, t, h, w).to(dtype)
        if kind == "at":
            kw = dict(tail_c=rn(n, cin, t, h, w).to(dtype), tail_sums_c=torch.zeros((cin, 2), dtype=torch.float64, device=dev))
        fused = lambda: ops.pw_bwd(gy, yraw, coef, dp, dx, dw, ops.EPI_ADD, x=x, add=add, **kw)
        wg = lambda: ops.pw_wgrad(gy, yraw, coef, x, dw)
        dg = lambda: ops.pw_dgrad(gy, yraw, coef, wt, dx, ops.EPI_ADD, add=add, w_panel=dp)
        by = 2 * n * t * h * w * (2 * cout + 3 * cin)
    keep = (gy, yraw, coef, wt, fp, dp, dx, dw, x, kw)
    return fused, wg, dg, by, keep


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def main():
    args = sys.argv[1:]
    dtype = torch.float16 if args[0] == "fp16" else torch.bfloat16
    sep = args.index("--") if "--" in args else len(args)
    layers, variants = args[1:sep], (args[sep + 1:] or [""])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    base_env = dict(os.environ)
    for spec in layers:
        f = spec.split(",")
        kind = f[0]
        n, cin, cout, t, h, w = map(int, f[1:7])
        fused, wg, dg, by, keep = build(kind, n, cin, cout, t, h, w, dtype, dev, g)
        res = {v: [] for v in variants}
        mode = {}
        for rnd in range(12):
            for v in variants:
                for k in list(os.environ):
                    if k.startswith("X3D_") and k not in base_env:
                        del os.environ[k]
                for kv in v.split():
                    k, val = kv.split("=")
                    os.environ[k] = val
                if fused() is not False:
                    mode[v] = "x3d_pw_bwd"
                    us = timed(fused)
                else:
                    mode[v] = "x3d_pw_wgrad + x3d_pw_dgrad"
                    wg(); dg()
                    us = timed(wg) + timed(dg)
                if rnd >= 2:
                    res[v].append(us)
        for v in variants:
            us = statistics.median(res[v])
            print(f"{spec:28s} {v or 'default':36s} {us:8.1f} us (min {min(res[v]):7.1f})  {by / us / 1e6:6.2f} TB/s  {mode[v]}", flush=True)


if __name__ == "__main__":
    main()
