"""Per-launch timing of one training step: every recorded launch of the plan is bracketed by HIP events
(3 repetitions, median), with the algorithmic bytes of the conv launches (SURVEY 8d convention) so each
kernel's distance from the HBM roofline is visible layer by layer.

    python tools/bench_layers.py [variant] [batch] [dtype]      (on the GPU box)
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x3d_tf_amd as x  # noqa: E402
from x3d_tf_amd.model import X3D  # noqa: E402

CLIP = {"XS": (4, 160), "S": (13, 160), "M": (16, 224), "L": (16, 312), "XL": (16, 312)}


def describe(name, st, eb):
    if st is None:
        return "", 0
    f = {k: getattr(st, k) for k, _ in st._fields_ if k in ("N", "C", "Cin", "Cout", "T", "H", "W", "stride", "epi")}
    if "T" not in f:
        return "", 0
    n, t, h, w = f["N"], f["T"], f["H"], f["W"]
    s = f.get("stride", 1) or 1
    ho, wo = -(-h // s), -(-w // s)
    if name == "x3d_pw_fwd":
        by = eb * n * t * (f["Cin"] * ho * wo + f["Cout"] * ho * wo)
        return f"{f['Cin']}->{f['Cout']} @{t}x{h}x{w} s{s}", by
    if name == "x3d_pw_dgrad":
        extra = {0: 0, 1: f["Cin"], 2: f["Cin"] // 4, 3: f["Cin"]}[f["epi"]]
        by = eb * n * t * h * w * (2 * f["Cout"] + f["Cin"] + extra)
        return f"{f['Cout']}->{f['Cin']} @{t}x{h}x{w} epi{f['epi']}", by
    if name == "x3d_pw_bwd":   # fused dgrad + wgrad: dY twice-read tensors once, conv input / braw once, dx written
        extra = {0: 0, 1: f["Cin"], 2: f["Cin"] // 4, 3: 0}[f["epi"]]
        rc = bool(getattr(st, "rc_panel", None))      # recomputed-output form: the conv's raw output is not read
        by = eb * n * t * h * w * ((1 if rc else 2) * f["Cout"] + 2 * f["Cin"] + extra)
        xs = " s2" if getattr(st, "x_stride", 0) == 2 else ""     # strided shortcut: g, the sampled input pixels, dx
        return f"{f['Cout']}<->{f['Cin']} @{t}x{h}x{w} epi{f['epi']}" + (" rc" if rc else "") + xs, by
    if name == "x3d_pw_wgrad":
        by = eb * n * t * ho * wo * (2 * f["Cout"] + f["Cin"])
        return f"{f['Cout']}x{f['Cin']} @{t}x{ho}x{wo}", by
    if name == "x3d_dw3d_fwd":
        by = eb * n * f["C"] * t * (h * w + ho * wo)
        return f"C{f['C']} @{t}x{h}x{w} s{s}", by
    if name == "x3d_dw3d_bwd":
        by = eb * n * f["C"] * t * (2 * h * w + 2 * ho * wo)
        return f"C{f['C']} @{t}x{h}x{w} s{s}", by
    return "", 0


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "M"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
    eb = 2 if dtype == torch.bfloat16 else 4
    cfg = x.get_config(variant)
    dev = torch.device("cuda:0")
    m = X3D(cfg, dtype=dtype, device=dev)
    t, s = CLIP[variant]
    clips = torch.randn(batch, t, s, s, 3, device=dev).to(dtype)
    labels = torch.randint(0, 400, (batch,), device=dev)
    for _ in range(2):
        pl = m.forward_backward(clips, labels)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    times = {}
    for rep in range(3):
        pl.zero_buf.zero_()
        m.flat_grads.zero_()
        for lname, lst in (("fwd", pl.fwd), ("bwd", pl.bwd)):
            evs = []
            for i, (name, fn, args) in enumerate(lst):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(*args, stream)
                e1.record()
                evs.append((i, name, e0, e1))
            torch.cuda.synchronize()
            for i, name, e0, e1 in evs:
                times.setdefault((lname, i, name), []).append(e0.elapsed_time(e1) * 1e3)
    rows = []
    for (lname, i, name), ts in times.items():
        lst = pl.fwd if lname == "fwd" else pl.bwd
        st = pl.structs.get((id(lst), i))
        desc, by = describe(name, st, eb)
        us = statistics.median(ts)
        rows.append((lname, i, name, desc, us, by))
    tot = sum(r[4] for r in rows)
    print(f"# {variant} B={batch} {dtype}: sum of launch medians {tot / 1e3:.2f} ms")
    agg = {}
    for r in rows:
        a = agg.setdefault(r[2], [0, 0.0, 0])
        a[0] += 1; a[1] += r[4]; a[2] += r[5]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        gbs = a[2] / (a[1] * 1e-6) / 1e9 if a[2] else 0
        print(f"{k:28s} n={a[0]:4d} {a[1] / 1e3:8.2f} ms  {gbs:8.1f} GB/s")
    print("# per launch (conv kernels)")
    for lname, i, name, desc, us, by in rows:
        if by:
            print(f"{lname} {i:4d} {name:14s} {desc:34s} {us:9.1f} us {by / 1e6:9.1f} MB {by / (us * 1e-6) / 1e9:8.1f} GB/s")
    print("# per launch (other kernels above 25 us)")
    for lname, i, name, desc, us, by in rows:
        if not by and us > 25.0:
            print(f"{lname} {i:4d} {name:24s} {us:9.1f} us")


if __name__ == "__main__":
    main()
