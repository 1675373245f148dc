#!/bin/bash
# Build a variant of libx3d_hip.so for A/B runs: tools/build_variant.sh NAME "extra hipcc flags" file.hip [file.hip ...]
# recompiles the named translation units with the extra flags and links them with the product's other objects into
# x3d-tf_amd/libx3d_hip_NAME.so; select it with X3D_HIP_LIB=x3d-tf_amd/libx3d_hip_NAME.so (tools only).
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; shift 2
python x3d-tf_amd/build.py > /dev/null
mkdir -p /tmp/x3d_variant_$name
objs=""
for o in x3d-tf_amd/csrc/obj/*.o; do
  b=$(basename $o .o); keep=1
  for f in "$@"; do [ "$(basename $f .hip)" == "$b" ] && keep=0; done
  [ $keep == 1 ] && objs="$objs $o"
done
for f in "$@"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -fno-slp-vectorize $extra -c x3d-tf_amd/csrc/$b.hip -o /tmp/x3d_variant_$name/$b.o &
done
wait
for f in "$@"; do objs="$objs /tmp/x3d_variant_$name/$(basename $f .hip).o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o x3d-tf_amd/libx3d_hip_$name.so $objs
echo x3d-tf_amd/libx3d_hip_$name.so
