# stride-2 backward on the small planes: flat 16-byte staging in the deep-prefetch kernel vs X3D_DW_PDFLAT=0
for shp in 216,16,28,28,2 432,16,14,14,2; do
  AB_ONLY=$shp python tools/ab_dw.py gpurun_out/pf1_$shp.json 64 || exit 1
  X3D_DW_PDFLAT=0 AB_ONLY=$shp python tools/ab_dw.py gpurun_out/pf0_$shp.json 64 || exit 1
  python tools/ab_dw.py --compare gpurun_out/pf0_$shp.json gpurun_out/pf1_$shp.json || exit 1
done
