#!/bin/bash
# same-box A/B of one plan option: bench.py alternating between the product and the option's historical switch
#   bash tools/ab_option.sh X3D_NO_SHORTCUT_COMPACT [runs=2] [extra bench args]
sw=$1; runs=${2:-2}; shift 2 2>/dev/null
for i in $(seq 1 $runs); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(f'product_$i        {d[\"value\"]:8.1f} clips/s  {d[\"ms_per_step\"]:.3f} ms/step')"
  env X3D_EXPERIMENTS=1 $sw=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(f'${sw}_$i {d[\"value\"]:8.1f} clips/s  {d[\"ms_per_step\"]:.3f} ms/step')"
done
