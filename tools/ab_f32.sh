for env in "X=0" "X3D_PW_F32_RESIDENT_KB=40" "X3D_PW_F32_RESIDENT_KB=40 X3D_PW_F32_KC=32" "X3D_PW_F32_RESIDENT_KB=24 X3D_PW_F32_KC=32" "X3D_PW_F32_RESIDENT_KB=160"; do
  echo "== $env"; env $env python tools/bench_layers.py S 32 fp32 2>/dev/null | sed -n 2,5p
done
