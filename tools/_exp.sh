python -m pytest tests/test_kernels_gpu.py -x -q -k "pw" 2>&1 | tail -4
python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -4
python tools/bench_layers.py M 64 > gpurun_out/exp_base.txt 2>&1
X3D_PW_TPBMIN=2 python tools/bench_layers.py M 64 > gpurun_out/exp_tpb2.txt 2>&1
X3D_PW_TPBMIN=1 python tools/bench_layers.py M 64 > gpurun_out/exp_tpb1.txt 2>&1
X3D_PW_TPBMIN=4 python tools/bench_layers.py M 64 > gpurun_out/exp_tpb4.txt 2>&1
head -2 gpurun_out/exp_*.txt
