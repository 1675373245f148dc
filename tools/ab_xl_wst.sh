#!/bin/bash
# XL pointwise shapes (60 clips of 16x312x312): resident-panel kernel (X3D_PW_WST_XL=0) against the weights-stationary shapes
SH="60,136,306,16,20,20,n,n 60,306,136,16,20,20,s,n 60,306,136,16,20,20,s,i 60,280,630,16,10,10,n,n 60,630,280,16,10,10,s,n 60,630,280,16,10,10,s,i 60,72,162,16,39,39,n,n 60,162,72,16,39,39,s,i 60,72,306,16,39,39,n,n 60,136,630,16,20,20,n,n"
for m in 0 1; do
  echo "== X3D_PW_WST_XL=$m"
  X3D_PW_WST_XL=$m python tools/pw_shape_bench.py fp16 $SH
done
