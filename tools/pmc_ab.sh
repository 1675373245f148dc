#!/bin/bash
# PMC passes over tools/ab_dw.py for ONE depthwise shape:  AB_ONLY=216,16,14,14,1 bash tools/pmc_ab.sh <tag> "<CTR ...>" ["<CTR ...>" ...]
# (export the kernel's A/B switches before calling).  Aggregated per kernel into gpurun_out/<tag>/pmc_<i>.csv.
set -u
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  rocprofv3 -M --pmc $grp --kernel-trace --output-format csv -d "$OUT/raw_$i" -- python3 "$REPO/tools/ab_dw.py" "$OUT/ab.json" 64 > "$OUT/pmc_$i.out" 2> "$OUT/pmc_$i.err" || { tail -5 "$OUT/pmc_$i.err"; exit 1; }
  python3 "$REPO/tools/pmc_sq.py" "$OUT/raw_$i" | grep -E "^kernel|dw3d_bwd" > "$OUT/pmc_$i.csv"
  rm -rf "$OUT/raw_$i"
  cat "$OUT/pmc_$i.csv"
  i=$((i+1))
done
