#!/bin/bash
# same-box A/B of the fused stem: bench.py alternating between the product and the two-kernel stem (plan option stem_fused off)
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(f'fused_$i      {d[\"value\"]:8.1f} clips/s  {d[\"ms_per_step\"]:.3f} ms/step')"
  X3D_EXPERIMENTS=1 X3D_NO_STEM_FUSED=1 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(f'two_kernel_$i {d[\"value\"]:8.1f} clips/s  {d[\"ms_per_step\"]:.3f} ms/step')"
done
