"""Per-launch timing of one INFERENCE forward (views of a few videos): every recorded launch of the plan bracketed by HIP
events (3 repetitions, median).

    python tools/bench_layers_infer.py [variant] [views] [T] [S] [dtype]      (on the GPU box)
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x3d_tf_amd as x  # noqa: E402
from x3d_tf_amd.model import X3D  # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "S"
    views = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    t = int(sys.argv[3]) if len(sys.argv) > 3 else 13
    s = int(sys.argv[4]) if len(sys.argv) > 4 else 182
    dtype = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[5] if len(sys.argv) > 5 else "fp16"]
    cfg = x.get_config(variant, ["TEST.NUM_TEMPORAL_VIEWS", views, "TEST.NUM_SPATIAL_CROPS", 1])
    dev = torch.device("cuda:0")
    m = X3D(cfg, dtype=dtype, device=dev)
    clips = torch.randn(views * 2, t, s, s, 3, device=dev).to(dtype)
    for _ in range(2):
        m(clips, training=False)
    torch.cuda.synchronize()
    pl = m._plan(views * 2, t, s, s, False)
    stream = torch.cuda.current_stream().cuda_stream
    times = {}
    for rep in range(3):
        evs = []
        for i, (name, fn, args) in enumerate(pl.fwd):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*args, stream)
            e1.record()
            evs.append((i, name, e0, e1))
        torch.cuda.synchronize()
        for i, name, e0, e1 in evs:
            times.setdefault((i, name), []).append(e0.elapsed_time(e1) * 1e3)
    rows = [(i, name, statistics.median(ts)) for (i, name), ts in times.items()]
    tot = sum(r[2] for r in rows)
    print(f"# {variant} {views * 2} clips of {t}x{s}x{s} {dtype}: sum of launch medians {tot / 1e3:.2f} ms")
    agg = {}
    for i, name, us in rows:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1; a[1] += us
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:28s} n={a[0]:4d} {a[1] / 1e3:8.2f} ms")
    print("# launches above 40 us")
    for i, name, us in rows:
        if us > 40.0:
            st = pl.structs.get((id(pl.fwd), i))
            desc = ""
            if st is not None:
                f = {k: getattr(st, k) for k, _ in st._fields_ if k in ("N", "C", "Cin", "Cout", "T", "H", "W", "stride")}
                desc = " ".join(f"{k}={v}" for k, v in f.items())
            print(f"{i:4d} {name:20s} {us:9.1f} us  {desc}")


if __name__ == "__main__":
    main()
