# stage-5 weight gradients (192 <-> 432 @7x7): eight-wave wide tile groups vs X3D_PW_WG_WIDE=0, with / without the atomic flush
for env in "X=0" "X3D_PW_WG_NOFLUSH=1" "X3D_PW_WG_WIDE=0" "X3D_PW_WG_WIDE=0 X3D_PW_WG_NOFLUSH=1"; do
  echo "== $env"
  env $env python tools/bench_layers.py M 64 2>/dev/null | grep -E "x3d_pw_wgrad +(192x432|432x192|216x96|96x216) @" | awk '{print $4, $5, $(NF-5), $(NF-4)}' | sort | uniq -c | awk '{print $2,$3,$4}' | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++} END{for(k in s) print k, s[k]/n[k]}'
done
