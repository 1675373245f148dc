"""MFMA utilisation of the pointwise-conv launches from a tools/bench_layers.py table:
achieved FLOP/s (2 * Cin * Cout * points per GEMM; the fused backward and the separate dgrad + wgrad are two GEMMs
per layer) over the dense bf16 MFMA peak of MI355X (2.5 PFLOP/s, MI355X_MICROARCH.md).  These GEMMs have K <= 432 and
move 2-10 bytes per MAC-row, so they sit on the HBM roofline, not the MFMA one: the number says how far.

    python tools/mfma_util.py profiles/r01m_per_launch_layers.txt [batch]
"""
import re
import sys

PEAK = 2.5e15


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    agg = {}
    for line in open(path):
        m = re.match(r"(fwd|bwd)\s+\d+\s+(x3d_pw_\w+)\s+(\d+)(->|<->|x)(\d+)\s+@(\d+)x(\d+)x(\d+)(?:\s+(s\d|epi\d))?\s+([\d.]+) us", line)
        if not m:
            continue
        _, name, a, _, b, t, h, w, extra, us = m.groups()
        a, b, t, h, w, us = int(a), int(b), int(t), int(h), int(w), float(us)
        if extra and extra.startswith("s") and extra != "s1":
            h, w = -(-h // 2), -(-w // 2)          # strided forward: output points
        gemms = 2 if name == "x3d_pw_bwd" else 1
        flops = 2.0 * a * b * n * t * h * w * gemms
        k = (name, f"{a}/{b} @{t}x{h}x{w}")
        g = agg.setdefault(k, [0, 0.0, 0.0])
        g[0] += 1; g[1] += us; g[2] += flops
    tot_f = sum(g[2] for g in agg.values()); tot_t = sum(g[1] for g in agg.values())
    print(f"# pointwise GEMMs of one train step: {tot_f / 1e12:.2f} TFLOP in {tot_t / 1e3:.2f} ms = "
          f"{tot_f / tot_t / 1e6:.1f} TFLOP/s = {100 * tot_f / (tot_t * 1e-6) / PEAK:.2f} % of the 2.5 PFLOP/s bf16 MFMA peak")
    for (name, shape), g in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tf = g[2] / (g[1] * 1e-6) / 1e12
        print(f"{name:14s} {shape:22s} n={g[0]:2d} {g[1]:8.1f} us  {tf:7.1f} TFLOP/s  {100 * tf * 1e12 / PEAK:5.2f} %")


if __name__ == "__main__":
    main()
