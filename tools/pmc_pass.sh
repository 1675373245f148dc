#!/bin/bash
# One PMC pass per counter group over two train steps:  bash tools/pmc_pass.sh <tag> "<CTR CTR ...>" ["<CTR ...>" ...]
# Aggregated per kernel with tools/pmc_sq.py into gpurun_out/<tag>/pmc_<i>.csv (raw per-dispatch CSVs are removed).
set -u
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  rocprofv3 -M --pmc $grp --kernel-trace --output-format csv -d "$OUT/raw_$i" -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.err" || { tail -5 "$OUT/pmc_$i.err"; exit 1; }
  python3 "$REPO/tools/pmc_sq.py" "$OUT/raw_$i" > "$OUT/pmc_$i.csv"
  rm -rf "$OUT/raw_$i"
  i=$((i+1))
done
ls -la "$OUT"
