#!/bin/bash
# Round 6 A/B: statistics producers adding into 8 copies (one per XCD) instead of 32, with and without
# the BatchNorm finalize folded into its consumers (plan option bn_fold).  Needs x3d-tf_amd/libx3d_hip_su8.so:
#   tools/build_variant.sh su8 "-DSTATS_USED=8" $(cd x3d-tf_amd/csrc && ls *.hip)
line() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(f'$1 {d[\"value\"]:8.1f} clips/s  {d[\"ms_per_step\"]:.3f} ms/step')"; }
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "copies32_finalize_launches_$i"
  X3D_EXPERIMENTS=1 X3D_BN_FOLD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "copies32_bn_fold_$i         "
  X3D_HIP_LIB=x3d-tf_amd/libx3d_hip_su8.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "copies8_finalize_launches_$i "
  X3D_HIP_LIB=x3d-tf_amd/libx3d_hip_su8.so X3D_EXPERIMENTS=1 X3D_BN_FOLD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "copies8_bn_fold_$i          "
done
