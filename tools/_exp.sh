python -m pytest tests/test_kernels_gpu.py -x -q -k "dw3d" 2>&1 | tail -3
python tools/bench_layers.py M 64 > gpurun_out/exp_l63.txt 2>&1
grep "dw3d_bwd  \|sum of" gpurun_out/exp_l63.txt | head -4
