python -m pytest tests/test_kernels_gpu.py -x -q -k stem 2>&1 | tail -5
python tools/bench_layers.py M 64 > gpurun_out/exp_stem.txt 2>&1
grep "stem\|sum of" gpurun_out/exp_stem.txt | head
