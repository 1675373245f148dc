#!/bin/bash
# matrix-core depthwise backward (14 x 14 and 28 x 28 planes): planes in flight PD = 2 (product) vs 4.
# Needs the experiments variant: tools/build_variant.sh exp "-DX3D_EXPERIMENTS" dw_mx.hip
export X3D_HIP_LIB=x3d-tf_amd/libx3d_hip_exp.so
for shp in 216,16,14,14,1 108,16,28,28,1; do
  AB_ONLY=$shp python tools/ab_dw.py gpurun_out/mxpd2_$shp.json 64 || exit 1
  X3D_DW_MX_PD=4 AB_ONLY=$shp python tools/ab_dw.py gpurun_out/mxpd4_$shp.json 64 || exit 1
  python tools/ab_dw.py --compare gpurun_out/mxpd2_$shp.json gpurun_out/mxpd4_$shp.json || exit 1
done
