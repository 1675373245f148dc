#!/bin/bash
# Timing experiments on the fused stem forward (results are WRONG in every arm but the first): which stage of a tick costs what.
#   for e in 1 2 4 8 6 15 16 32 48; do tools/build_variant.sh sfexp$e "-DSF_EXP=$e" stem_fused.hip; done      (build host)
#   bash tools/ab_stem_parts.sh                                                                        (GPU box)
for e in ${SF_ARMS:-0 1 2 4 8 6 15 16 32 48}; do
  if [ $e = 0 ]; then lib=x3d-tf_amd/libx3d_hip.so; else lib=x3d-tf_amd/libx3d_hip_sfexp$e.so; fi
  echo "SF_EXP=$e $(X3D_HIP_LIB=$lib python tools/stem_bench.py 2>/dev/null | grep x3d_stem_fwd)"
done
