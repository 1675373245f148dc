"""Depthwise HBM GB/s from ROCPROF durations (the second half of BASELINE's metric): joins the rocprofv3 kernel-stats table
of a bench.py run (`-M --kernel-trace --stats`: mangled names, AverageNs) with the algorithmic bytes per launch bench.py
reports for every depthwise instantiation (forward e*(X + Y), fused backward e*(X + dY + dX)).

    python tools/dw_gbs.py <kernel_stats.csv> <bench_line.json> [pmc_traffic.json]
"""
import csv
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import canonical  # noqa: E402


def main():
    stats = {}
    with open(sys.argv[1], newline="") as fh:
        for row in csv.DictReader(fh):
            stats[canonical(row["Name"])] = row
    line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    pmc = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
    tot_b = tot_t = 0.0
    print(f"{'kernel':52s} {'calls':>5s} {'rocprof avg us':>14s} {'bench avg us':>12s} {'alg. GB':>8s} {'GB/s':>7s} {'of 8 TB/s':>9s} {'PMC traffic / alg.':>18s}")
    for k in line["kernels"]:
        st = stats.get(k["kernel"])
        if st is None:
            print(f"{k['kernel']:52s}  (not in the rocprof table)")
            continue
        us = float(st["AverageNs"]) / 1e3
        by = k["algorithmic_bytes_per_launch"]
        gbs = by / us / 1e3
        tr = pmc.get(k["kernel"], {}).get("traffic_bytes_per_launch")
        print(f"{k['kernel']:52s} {int(st['Calls']):5d} {us:14.1f} {k['avg_us']:12.1f} {by / 1e9:8.3f} {gbs:7.0f} {gbs / 8000:9.3f} "
              f"{(tr / by if tr else float('nan')):18.2f}")
        n = k["launches"] / 2          # (bench.py collects the `kernels` table over TWO steps after the timed region, not over `steps`)
        tot_b += by * n
        tot_t += us * n
    print(f"# all depthwise launches of one step: {tot_b / 1e9:.2f} GB algorithmic in {tot_t / 1e3:.2f} ms = {tot_b / tot_t / 1e3:.0f} GB/s "
          f"({tot_b / tot_t / 1e3 / 8000:.3f} of the 8 TB/s HBM peak)")


if __name__ == "__main__":
    main()
