python -m pytest tests/test_kernels_gpu.py -x -q -k pw 2>&1 | tail -3
python tools/bench_layers.py M 64 > gpurun_out/exp_gb.txt 2>&1
head -2 gpurun_out/exp_gb.txt
