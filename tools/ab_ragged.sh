# ragged planes against their aligned neighbours (A/B: X3D_DW_FLAT=0 = run-time-width path, X3D_DW_PDFLAT=0 = no flat deep-prefetch)
for shp in 216,16,39,39,2 216,16,40,40,2 108,16,78,78,2 108,16,80,80,2 54,16,156,156,2; do
  AB_ONLY=$shp python tools/ab_dw.py gpurun_out/ab_$shp.json 16 || exit 1
done
