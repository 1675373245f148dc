#!/bin/bash
# H- and W-tiled matrix-core depthwise backward (dw_mxg.hip) against the vector kernels it replaces (X3D_DW_MXG=0):
# X3D-L planes (39 x 39, 78 x 78 at batch 16) and the X3D-M 56 x 56 plane (batch 64).
# Needs the experiments variant: tools/build_variant.sh exp "-DX3D_EXPERIMENTS" dw_mxg.hip
export X3D_HIP_LIB=x3d-tf_amd/libx3d_hip_exp.so
for spec in 108,16,39,39,1:16 54,16,78,78,1:16 54,16,56,56,1:64 162,16,39,39,1:8 72,16,78,78,1:8; do
  shp=${spec%%:*}; nb=${spec##*:}
  X3D_DW_MXG=0 AB_ONLY=$shp python tools/ab_dw.py gpurun_out/mxg0_$shp.json $nb > /dev/null || exit 1
  AB_ONLY=$shp python tools/ab_dw.py gpurun_out/mxg1_$shp.json $nb > /dev/null || exit 1
  python tools/ab_dw.py --compare gpurun_out/mxg0_$shp.json gpurun_out/mxg1_$shp.json
done
exit 0
