python -m pytest tests/test_kernels_gpu.py -x -q -k "pw_fwd or pw_wgrad" 2>&1 | tail -4
python tools/bench_layers.py M 64 > gpurun_out/exp_str.txt 2>&1
grep "^x3d\|sum of" gpurun_out/exp_str.txt | head -3
