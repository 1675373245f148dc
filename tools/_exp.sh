python -m pytest tests/test_kernels_gpu.py -x -q -k "se_" 2>&1 | tail -5
python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
python tools/bench_layers.py M 64 > gpurun_out/exp_se.txt 2>&1
grep "se_\|sum of" gpurun_out/exp_se.txt | head
