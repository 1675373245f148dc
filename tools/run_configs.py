"""Runs the other BASELINE.json configurations end to end on one MI355X and reports throughput
(they are parity-test cases, not bench lines): S fp32 B=32 train, L bf16 train, XL 30-view inference in fp16 (the
reference's mixed_float16) and bf16."""
import gc, os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x3d_tf_amd as x
from x3d_tf_amd.model import X3D
from x3d_tf_amd.train import Trainer

def train_rate(variant, batch, t, s, dtype, steps=5):
    cfg = x.get_config(variant)
    gc.collect(); torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()          # the footprint of THIS configuration, not the largest one so far
    m = X3D(cfg, dtype=dtype, device="cuda:0")
    tr = Trainer(m, cfg)
    clips = torch.randn(batch, t, s, s, 3, device="cuda").to(dtype)
    labels = torch.randint(0, 400, (batch,), device="cuda")
    for _ in range(2):
        pl = tr.step(clips, labels, 0.01)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        pl = tr.step(clips, labels, 0.01)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    loss = float(tr.loss(pl).item())
    mem = torch.cuda.max_memory_allocated() / 1e9
    del m, tr
    torch.cuda.empty_cache()
    return dict(variant=variant, mode="train", dtype=str(dtype), batch=batch, clip=f"{t}x{s}x{s}", clips_per_s=steps * batch / el,
                ms_per_step=1e3 * el / steps, loss=loss, mem_GB=mem)

def infer_rate(variant, videos, views, crops, t, s, dtype, steps=5):
    cfg = x.get_config(variant, ["TEST.NUM_TEMPORAL_VIEWS", views, "TEST.NUM_SPATIAL_CROPS", crops])
    gc.collect(); torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    m = X3D(cfg, dtype=dtype, device="cuda:0")
    n = videos * views * crops
    clips = torch.randn(n, t, s, s, 3, device="cuda").to(dtype)
    for _ in range(2):
        out = m(clips, training=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        out = m(clips, training=False)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    ok = bool(torch.isfinite(out).all()) and tuple(out.shape) == (videos, 400)
    mem = torch.cuda.max_memory_allocated() / 1e9
    del m
    torch.cuda.empty_cache()
    return dict(variant=variant, mode="inference", dtype=str(dtype), videos=videos, views=views * crops, clip=f"{t}x{s}x{s}",
                clips_per_s=steps * n / el, videos_per_s=steps * videos / el, ms_per_batch=1e3 * el / steps, ok=ok, mem_GB=mem)

def cpu_infer_rate(variant, videos, views, crops, t, s, steps=5):
    """BASELINE config 1 as it is stated: the forward pass on the CPU (the reference's TF CPU path is not runnable here: the
    CPU oracle -- PyTorch-CPU fp32 restatement of the reference graph -- on the host cores), one 10-view video at a time."""
    from oracle import x3d_oracle as O
    from x3d_tf_amd.params import init_params
    cfg = x.get_config(variant, ["TEST.NUM_TEMPORAL_VIEWS", views, "TEST.NUM_SPATIAL_CROPS", crops])
    arch = x.build_arch(cfg)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    p = init_params(arch, seed=0)
    n = videos * views * crops
    clips = torch.randn(n, t, s, s, 3)
    with torch.no_grad():
        out = O.forward(p, clips, arch, training=False)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = O.forward(p, clips, arch, training=False)
        el = time.perf_counter() - t0
    return dict(variant=variant, mode="inference (CPU oracle, kind=port)", dtype="torch.float32", videos=videos, views=views * crops,
                clip=f"{t}x{s}x{s}", clips_per_s=steps * n / el, videos_per_s=steps * videos / el, ms_per_batch=1e3 * el / steps,
                cores=cores, ok=bool(torch.isfinite(out).all()) and tuple(out.shape) == (videos, 400))


CONFIGS = {   # name: (function, arguments)
    "cfg2": (train_rate, ("S", 32, 13, 160, torch.float32)),            # BASELINE config 2
    "cfg4": (train_rate, ("L", 16, 16, 312, torch.bfloat16)),           # config 4 (yaml batch 16)
    "xl_train": (train_rate, ("XL", 8, 16, 312, torch.bfloat16)),
    "cfg5": (infer_rate, ("XL", 2, 10, 3, 16, 312, torch.float16)),     # config 5: 30 views / video, fp16 (Keras mixed_float16)
    "cfg5_bf16": (infer_rate, ("XL", 2, 10, 3, 16, 312, torch.bfloat16)),
    "m_fp16": (train_rate, ("M", 64, 16, 224, torch.float16)),          # the headline workload in fp16 with loss scaling
    "cfg1": (infer_rate, ("XS", 8, 10, 1, 4, 160, torch.float32)),      # config 1 on the GPU
    "cfg1_cpu": (cpu_infer_rate, ("XS", 1, 10, 1, 4, 160)),              # config 1 as stated: forward on the host cores
    "s_bf16": (train_rate, ("S", 64, 13, 160, torch.bfloat16)),         # config 2's model in 16-bit storage: 13 frames -> ragged rows (P % 8 != 0)
    "xs_bf16": (train_rate, ("XS", 64, 4, 160, torch.bfloat16)),
}

if __name__ == "__main__":
    # python tools/run_configs.py [--steps K] [name ...]      (no names: all of them)
    argv = sys.argv[1:]
    steps = 5
    if argv[:1] == ["--steps"]:
        steps, argv = int(argv[1]), argv[2:]
    for name in (argv or list(CONFIGS)):
        fn, args = CONFIGS[name]
        r = fn(*args, steps=steps)
        r["name"] = name
        print(json.dumps(r), flush=True)
