python -m pytest tests/test_kernels_gpu.py -x -q -k "dw3d" 2>&1 | tail -3
python tools/bench_layers.py M 64 > gpurun_out/exp_dw2.txt 2>&1
grep "^x3d\|sum of" gpurun_out/exp_dw2.txt | head -3
