#!/bin/bash
# Round-end measurement on the GPU box:  bash tools/profile_round.sh <tag> [pmc] [core|configs]
#   (third argument: only the X3D-M part / only the other configurations -- each fits one 20-minute gpurun call)
#   bench line, rocprofv3 kernel-trace stats of the same command, per-launch layer timing and (with "pmc")
#   two separate PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) for HBM traffic.
# Everything lands in gpurun_out/<tag>/; copy the summaries worth judging into profiles/.
set -u
TAG=${1:-rXX}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PART=${3:-all}
if [ "$PART" != "configs" ]; then
python3 "$REPO/bench.py" --steps 10 --warmup 3 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench_line.json"
rocprofv3 -M --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$REPO/bench.py" --steps 8 --warmup 3 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err"
find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
python3 "$REPO/tools/bench_layers.py" M 64 > "$OUT/per_launch_layers.txt" 2>&1
if [ "${2:-}" = "pmc" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 -M --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_$ctr" -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$ctr.json" 2> "$OUT/pmc_$ctr.err"
  done
  python3 "$REPO/tools/pmc_traffic.py" "$OUT" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
  # the raw per-dispatch CSVs are large; keep only the summary
  rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
fi
rm -rf "$OUT/prof"
python3 "$REPO/tools/dw_gbs.py" "$OUT/kernel_stats.csv" "$OUT/bench_line.json" "$OUT/pmc_traffic.json" > "$OUT/dw_gbs.txt" 2>&1
python3 "$REPO/tools/mfma_util.py" "$OUT/per_launch_layers.txt" > "$OUT/mfma_util.txt" 2>&1
fi
if [ "$PART" = "core" ]; then ls -la "$OUT"; exit 0; fi
# the other BASELINE configurations: throughput lines, and one rocprofv3 kernel-stats table each for configs 2, 4 and 5
python3 "$REPO/tools/run_configs.py" > "$OUT/other_configs.jsonl" 2> "$OUT/other_configs.err"
for c in cfg2 cfg4 cfg5; do
  rocprofv3 -M --kernel-trace --stats --output-format csv -d "$OUT/prof_$c" -- python3 "$REPO/tools/run_configs.py" --steps 2 $c > "$OUT/prof_$c.json" 2> "$OUT/prof_$c.err"
  find "$OUT/prof_$c" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$c.csv" \;
  rm -rf "$OUT/prof_$c"
done
python3 "$REPO/tools/bench_layers_infer.py" XL 30 16 312 fp16 > "$OUT/per_launch_layers_XL_infer.txt" 2>&1
ls -la "$OUT"
