#!/bin/bash
# Timing experiments on the fp32 pipelined pointwise kernel (pw_gemm_f32p.h; results are WRONG in every arm but the first): exact
# kernel durations from a rocprofv3 kernel trace per arm.
#   for e in 1 4 8 16 32 64; do tools/build_variant.sh f32pexp$e "-DF32P_EXP=$e" pw_fwd.hip pw_dgrad.hip; done     (build host)
#   bash tools/ab_f32r_parts.sh                                                                                   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in ${F32P_ARMS:-0 1 4 8 16 32 64}; do
  if [ $e = 0 ]; then lib=$PWD/x3d-tf_amd/libx3d_hip.so; else lib=$PWD/x3d-tf_amd/libx3d_hip_f32pexp$e.so; fi
  rm -rf gpurun_out/f32tr_$e
  X3D_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace -M --output-format csv -d gpurun_out/f32tr_$e -o t -- python tools/bench_f32r.py > gpurun_out/f32tr_run_$e.txt 2>&1
  echo "== F32P_EXP=$e"; python tools/f32_trace.py gpurun_out/f32tr_$e
done
