"""rocprofv3 kernel trace (csv) of a few train steps: the collective kernels (name contains nccl / rccl), their durations, and the
idle gaps of the device timeline above a threshold -- what a one-rank rehearsal of the data-parallel step costs and where.

    rocprofv3 -M --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/trace_gaps.py DIR [gap_us=15]
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    t0 = rows[0][0]
    coll = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
    print(f"{len(rows)} dispatches, {len(coll)} collective kernels")
    for s, e, n, q in coll[-16:]:
        # what compute kernels overlap it
        ov = [r[2][:40] for r in rows if r[2] != n and r[0] < e and r[1] > s]
        print(f"  t={((s - t0) / 1e3):10.1f} us  dur {((e - s) / 1e3):7.1f} us  queue {q}  {n[:60]}  overlaps {len(ov)}: {ov[:3]}")
    # busy union of ALL kernels, gaps above the threshold in the last third of the trace (steady state)
    cut = rows[len(rows) * 2 // 3][0]
    end = 0
    gaps = []
    for s, e, n, q in rows:
        if s > end and end and s >= cut and (s - end) / 1e3 > thr:
            gaps.append(((s - end) / 1e3, n[:50], s))
        end = max(end, e)
    tot = sum(g[0] for g in gaps)
    print(f"gaps > {thr} us in the last third: {len(gaps)}, {tot:.0f} us in all")
    for g, n, s in sorted(gaps, reverse=True)[:12]:
        print(f"  {g:8.1f} us before {n}")


if __name__ == "__main__":
    main()
