python -m pytest tests/test_kernels_gpu.py -x -q -k "dw3d" 2>&1 | tail -2
python tools/bench_layers.py M 64 > gpurun_out/exp_ldsv.txt 2>&1
grep "sum of" gpurun_out/exp_ldsv.txt
