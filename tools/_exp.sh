python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -4
python tools/bench_layers.py M 64 > gpurun_out/exp_dpp.txt 2>&1
head -2 gpurun_out/exp_dpp.txt
