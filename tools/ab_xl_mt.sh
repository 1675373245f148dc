#!/bin/bash
# XL stage-4 / stage-5 pointwise shapes (60 clips of 16x312x312) under different panel heights of the resident-panel kernel
SH="60,136,306,16,20,20,n,n 60,306,136,16,20,20,s,n 60,306,136,16,20,20,s,i 60,280,630,16,10,10,n,n 60,630,280,16,10,10,s,n 60,630,280,16,10,10,s,i 60,72,162,16,39,39,n,n 60,162,72,16,39,39,s,i 60,32,72,16,78,78,n,n 60,72,32,16,78,78,s,i"
for m in "" "9:3,5:3,10:4,20:4" "9:3,5:7,10:7,20:7" "9:2,5:4,10:3,20:3"; do
  echo "== X3D_PW_MTMAP=$m"
  X3D_PW_MTMAP="$m" python tools/pw_shape_bench.py fp16 $SH
done
echo "== X3D_PW_WS=1 (weights streamed wherever legal)"
X3D_PW_WS=1 python tools/pw_shape_bench.py fp16 $SH
