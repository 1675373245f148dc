source tools/gpu_steps.sh
mkdir -p gpurun_out
run 900 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py -k "full_size or recomputed_output or (train_step_fp32 and S-8)" -q -s -p no:cacheprovider > gpurun_out/r05_newtests.log 2>&1
echo "newtests done"; tail -5 gpurun_out/r05_newtests.log
run 300 python -m pytest tests/test_dist.py -m gpu -q -p no:cacheprovider > gpurun_out/r05_dist.log 2>&1
tail -3 gpurun_out/r05_dist.log
run 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench0.json 2> gpurun_out/r05_bench0.err
cat gpurun_out/r05_bench0.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_us'], d['roofline_model']['frac'])"
run 600 python tools/seed_sweep.py > gpurun_out/r05_seed_sweep.log 2>&1
tail -30 gpurun_out/r05_seed_sweep.log
