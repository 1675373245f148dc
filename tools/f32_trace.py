"""Exact kernel durations of tools/bench_f32r.py's layers from a rocprofv3 kernel trace (the tool's own event timing includes the
host's launch latency):  rocprofv3 --kernel-trace -M --output-format csv -d gpurun_out/f32tr -o t -- python tools/bench_f32r.py ;
python tools/f32_trace.py gpurun_out/f32tr"""
import csv, glob, os, statistics, sys
root = sys.argv[1]
f = [p for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "pw_f32" in r["Kernel_Name"] or "pw_gemm_kernel" in r["Kernel_Name"]]
for i in range(0, len(ks), 8):
    grp = ks[i:i + 8]
    print(f"{grp[0][0][:70]:70s} n={len(grp)} median {statistics.median(t for _, t in grp):7.1f} us")
