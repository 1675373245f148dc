"""Benchmark of the X3D hot path on MI355X: clips/sec of the fwd + bwd (+ SGD) train step.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N > 1 launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``, one rank per GPU over
RCCL).  W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides,
max over ranks; rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): X3D-M, clips of 16x224x224, batch
64 per GPU, bf16 activation storage (fp32 accumulation; matrix-core operands -- the pointwise convs and the depthwise
planes dw_mx.hip covers -- rounded to bf16), synthetic N(0,1) clips (NTHWC at the module
boundary, already resident in HBM), random-init weights; weak scaling (per-GPU batch fixed).

Extra objects on the JSON line:
  roofline      dominant depthwise kernel instantiation: algorithmic bytes per launch (SURVEY 8d:
                fwd e*(X + Y), fused bwd e*(X + dY + dX)) / average launch duration measured with HIP
                events on the launch stream inside the timed region; peak 8 TB/s.  Only THAT instantiation's
                launches (two per step) carry events in the timed region: which one dominates is found in two
                untimed probe steps after the warm-up.
  kernels       the same figure for every depthwise instantiation launched, from two steps AFTER the timed region
                (52 event pairs per step stay out of `value`).
  roofline_model  BASELINE.md's whole-step figure: clips/s * 1.262 GB / 8 TB/s.
  mfma_util     MFMA utilisation of the pointwise convs (north_star's second figure): their FLOPs over their HIP-event
                time in three extra steps after the timed region, against the dense 16-bit matrix-core peak.
  cpu_baseline  the CPU oracle (a PyTorch-CPU restatement of the reference graph; TensorFlow is not
                available) timed on the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver: must be set before HIP initialises

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import x3d_tf_amd as x3d  # noqa: E402
from x3d_tf_amd import arch as A  # noqa: E402
from x3d_tf_amd import dist as xdist  # noqa: E402

CLIP = {"XS": (4, 160), "S": (13, 160), "M": (16, 224), "L": (16, 312), "XL": (16, 312)}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 matrix-core peak (MI355X_MICROARCH.md; the sparse headline figure is 2x this)


def dw_key(pl, lst, i):
    """Name of the kernel instantiation this recorded depthwise launch runs -- asked of the library's own
    dispatch (x3d_dw3d_kernel_name), so it is the kernel the rocprofv3 summary lists."""
    from x3d_tf_amd import hip
    return hip.dw3d_kernel_name(pl.structs[(id(lst), i)])


class KernelTimer:
    """HIP events around selected launches of a plan (the launches run on torch's current stream, so
    torch.cuda.Event records on the stream the kernels are launched on)."""

    def __init__(self, model, pl, elem_bytes):
        self.slots = []   # (list_name, index, key, algorithmic bytes)
        for lname, lst, field in (("fwd", pl.fwd, "sb"), ("bwd", pl.bwd, None)):
            for i, (name, fn, args) in enumerate(lst):
                if name in ("x3d_dw3d_fwd", "x3d_dw3d_bwd"):
                    self.slots.append([lname, i, name, None, None])
        # shapes in launch order: forward walks blocks first->last, backward last->first
        fwd_slots = [s for s in self.slots if s[0] == "fwd"]
        bwd_slots = [s for s in self.slots if s[0] == "bwd"]
        for s, B in zip(fwd_slots, pl.blocks):
            b = B.spec
            x_el = pl.n * b.inner * pl.t * B.hh * B.ww
            y_el = pl.n * b.inner * pl.t * B.ho * B.wo
            s[3], s[4] = dw_key(pl, pl.fwd, s[1]), elem_bytes * (x_el + y_el)
        for s, B in zip(bwd_slots, reversed(pl.blocks)):
            b = B.spec
            x_el = pl.n * b.inner * pl.t * B.hh * B.ww
            y_el = pl.n * b.inner * pl.t * B.ho * B.wo
            s[3], s[4] = dw_key(pl, pl.bwd, s[1]), elem_bytes * (2 * x_el + y_el)
        self.events = []

    def wrap(self, pl):
        """Monkey-patch the plan's run() so the selected launches are bracketed by events."""
        marks = {(s[0], s[1]): s for s in self.slots}
        timer = self
        orig_fwd, orig_bwd = pl.fwd, pl.bwd

        def run(lst, start=0, stop=None):
            lname = "fwd" if lst is orig_fwd else "bwd"
            stream = torch.cuda.current_stream().cuda_stream
            stop_ = len(lst) if stop is None else stop
            for i in range(start, stop_):
                name, fn, args = lst[i]
                slot = marks.get((lname, i)) if timer.enabled else None
                if slot is not None and (timer.only is None or slot[3] in timer.only):
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    st = fn(*args, stream)
                    e1.record()
                    timer.events.append((slot, e0, e1))
                else:
                    st = fn(*args, stream)
                if st != 0:
                    from x3d_tf_amd import hip
                    hip.check(st, name)
        self.enabled = False
        self.only = None      # None: every depthwise launch; a set of instantiation names: those launches only
        pl.run = run

    def reset(self):
        self.events = []

    def summary(self):
        agg = {}
        for slot, e0, e1 in self.events:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(slot[3], dict(launches=0, ms=0.0, bytes=0))
            a["launches"] += 1
            a["ms"] += ms
            a["bytes"] += slot[4]
        out = []
        for k, a in agg.items():
            out.append(dict(kernel=k, launches=a["launches"], avg_us=1e3 * a["ms"] / a["launches"],
                            total_ms=a["ms"], algorithmic_bytes_per_launch=a["bytes"] / a["launches"],
                            achieved_GBs=a["bytes"] / (a["ms"] * 1e-3) / 1e9))
        out.sort(key=lambda d: -d["total_ms"])
        return out


PW_ENTRIES = ("x3d_pw_fwd", "x3d_pw_dgrad", "x3d_pw_wgrad", "x3d_pw_bwd")


def mfma_utilisation(model, pl, trainer, clips, labels, lr, steps=3):
    """north_star's second figure: MFMA utilisation of the pointwise convs.  Every pointwise launch of `steps` extra steps
    (AFTER the timed region) is bracketed by HIP events on the launch stream; FLOPs = 2 * Cin * Cout * output points per
    GEMM (the fused backward x3d_pw_bwd is two GEMMs), over the dense 16-bit matrix-core peak."""
    slots = {}
    for lname, lst in (("fwd", pl.fwd), ("bwd", pl.bwd)):
        for i, item in enumerate(lst):
            if item is None or item[0] not in PW_ENTRIES:
                continue
            if (id(lst), i) in pl.side_entries:   # forked onto the side stream (X3D_SIDE_WGRAD=1): events on this stream
                continue                          # would bracket nothing and count the FLOPs against ~zero time
            st = pl.structs[(id(lst), i)]
            s_ = getattr(st, "stride", 1) or 1
            ho, wo = -(-st.H // s_), -(-st.W // s_)
            gemms = 2 if item[0] == "x3d_pw_bwd" else 1
            slots[(lname, i)] = 2.0 * st.Cin * st.Cout * st.N * st.T * ho * wo * gemms
    events = []
    orig_fwd = pl.fwd
    prev_run = pl.run

    def run(lst, start=0, stop=None):
        lname = "fwd" if lst is orig_fwd else "bwd"
        stream = torch.cuda.current_stream().cuda_stream
        for i in range(start, len(lst) if stop is None else stop):
            name, fn, args = lst[i]
            fl = slots.get((lname, i))
            if fl is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args, stream)
                e1.record()
                events.append((fl, e0, e1))
            else:
                rc = fn(*args, stream)
            if rc != 0:
                from x3d_tf_amd import hip
                hip.check(rc, name)
    pl.run = run
    try:
        for _ in range(steps):
            trainer.step(clips, labels, lr)
        torch.cuda.synchronize()
    finally:
        pl.run = prev_run
    ms = sum(e0.elapsed_time(e1) for _, e0, e1 in events)
    flops = sum(fl for fl, _, _ in events)
    tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {"kernels": "pointwise convs (x3d_pw_fwd / dgrad / wgrad / bwd): %d launches per step" % (len(events) // max(steps, 1)),
            "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
            "flop_per_step": flops / steps, "ms_per_step": ms / steps,
            "note": "K, N <= 432 GEMMs moving 2-4 bytes per 2*K flops: HBM-bound by construction, the figure is reported, not optimised for"}


def cpu_baseline(variant, seconds_budget=25.0, batch=1):
    """fwd + bwd of the CPU oracle on `batch` clips of the benchmark shape, all host cores."""
    from oracle import x3d_oracle as O
    from x3d_tf_amd.params import init_params
    cfg = x3d.get_config(variant)
    arch = x3d.build_arch(cfg)
    t, s = CLIP[variant]
    # PyTorch-CPU collapses when oversubscribed (measured: 256 threads on the GPU box's host = 190 s/clip,
    # 8 threads in the build container = 1.7 s/clip), so the thread count is capped and reported.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    p = init_params(arch, seed=0)
    torch.manual_seed(0)
    xin = torch.randn(batch, t, s, s, 3)
    labels = torch.randint(0, arch.num_classes, (batch,))
    t0 = time.perf_counter()
    O.train_step(p, xin, labels, arch, lr=0.01, apply_update=True)        # warm-up
    warm = time.perf_counter() - t0
    done, el = 0, 0.0
    t0 = time.perf_counter()
    while warm < seconds_budget / 2:          # a host this slow is timed on the warm-up step alone
        O.train_step(p, xin, labels, arch, lr=0.01, apply_update=True)
        done += 1
        el = time.perf_counter() - t0
        if el + warm > seconds_budget or done >= 30:   # about 10-25 s of CPU work on the GPU box's host
            break
    steps_txt = f"1 warm-up + {done} timed steps" if done else "the single (warm-up) step"
    value = done * batch / el if done else batch / warm
    return dict(value=value, unit="clips/s", cores=cores, kind="port",
                sample=f"CPU restatement of the reference graph (PyTorch-CPU fp32 oracle; TensorFlow unavailable): "
                       f"X3D-{variant} fwd+bwd+SGD, batch {batch} of {t}x{s}x{s}, {steps_txt}, {cores} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--variant", default="M")
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="activation storage: bf16 (the headline), fp16 (the reference's mixed_float16 mode; dynamic loss scale), fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Invoked directly with --gpus N: start the N ranks as CHILD processes (torchrun, one rank per GPU) and exit with
        # their status.  This process never loads the HIP runtime: a process that initialised the GPU must not be replaced
        # or forked, and torch.cuda.device_count() is NOT safe here (without amdsmi it falls through to hipGetDeviceCount,
        # which opens /dev/kfd) -- the GPUs are counted from sysfs instead.  When the topology cannot be read (None) the
        # ranks make the check themselves (xdist.local_device raises for a LOCAL_RANK without a GPU).
        have = xdist.visible_gpu_count()
        if have is not None and have < args.gpus and os.environ.get("X3D_DIST_BACKEND") != "gloo":
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node (one process per GPU); "
                             "refusing to time fewer GPUs than asked for")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    rank, local_rank, world = xdist.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` directly, which starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs one MI355X GPU per rank and none is visible (no CPU fallback for the hot path)")
    device = torch.device(f"cuda:{xdist.local_device(local_rank)}")
    torch.cuda.set_device(device)

    from x3d_tf_amd.model import X3D
    from x3d_tf_amd.train import Trainer
    cfg = x3d.get_config(args.variant)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    model = X3D(cfg, dtype=dtype, device=device, seed=0)
    trainer = Trainer(model, cfg)
    t, s = CLIP[args.variant]
    B = args.batch
    g = torch.Generator(device=device)
    g.manual_seed(1000 + rank)
    clips = torch.randn((B, t, s, s, 3), generator=g, device=device, dtype=torch.float32).to(dtype)
    labels = torch.randint(0, model.num_classes, (B,), generator=g, device=device)
    lr = cfg.TRAIN.WARMUP_LR

    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    for _ in range(max(args.warmup, 1)):
        pl = trainer.step(clips, labels, lr)
    torch.cuda.synchronize()
    trainer.reducer.exposed_ms()      # (drop the warm-up steps' measurements)
    timer = KernelTimer(model, pl, 4 if dtype == torch.float32 else 2)
    timer.wrap(pl)
    timer.enabled = True
    for _ in range(2):                # untimed probe: which depthwise instantiation has the largest total
        pl = trainer.step(clips, labels, lr)
    torch.cuda.synchronize()
    probe = timer.summary()
    timer.reset()
    timer.only = {probe[0]["kernel"]} if probe else set()
    trainer.reducer.exposed_ms()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pl = trainer.step(clips, labels, lr)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = xdist.gather_over_ranks(elapsed, device)      # every rank's own clock around the same K steps
    elapsed = max(per_rank)
    loss = float(trainer.loss(pl).item())
    coll = trainer.collective_stats()     # (exposed_ms over the timed steps only)
    dominant = timer.summary()        # the dominant instantiation, HIP events inside the timed region
    timer.reset()
    timer.only = None
    for _ in range(2):                # every depthwise instantiation, after the timed region
        trainer.step(clips, labels, lr)
    torch.cuda.synchronize()
    timer.enabled = False
    mfma = mfma_utilisation(model, pl, trainer, clips, labels, lr) if (world == 1 and rank == 0) else None
    if world == 1 and trainer.collectives:
        # one-rank rehearsal (X3D_DIST_REHEARSE=1): the same K steps again WITHOUT hooks and collectives -- what the data-parallel
        # machinery costs a step before a byte crosses xGMI (profiles/r06_collective_overhead.txt: it is the first cross-queue
        # dependency of the step in the HIP runtime, ~0.2 ms, not RCCL work)
        trainer.collectives = trainer.reducer.active = trainer.sync_moving_stats = False
        for _ in range(2):
            trainer.step(clips, labels, lr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(clips, labels, lr)
        torch.cuda.synchronize()
        coll["overhead_ms_vs_no_collectives"] = 1e3 * (elapsed - (time.perf_counter() - t1)) / args.steps

    if rank == 0:
        clips_s = args.steps * B * world / elapsed
        kernels = timer.summary()
        dom = dominant[0] if dominant else None
        w = A.workload(model.arch, t, s, s)
        eb = 4 if dtype == torch.float32 else 2
        step_bytes_per_clip = 3 * w["total_elements"] * eb
        # HBM traffic per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.py:
        # separate FETCH_SIZE / WRITE_SIZE runs, (2*FETCH + WRITE) * 1024 on gfx950); null if not collected
        # for this exact instantiation.
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if dom is not None and os.path.exists(pmc_file) and args.variant == "M" and B == 64:
            with open(pmc_file) as fh:
                rec = json.load(fh).get(dom["kernel"])
            if rec and rec.get("dtype", args.dtype) == args.dtype:
                traffic = rec["traffic_bytes_per_launch"]
        out = {
            "metric": "clips/sec (fwd+bwd) X3D-%s %dx%d^2; depthwise HBM GB/s" % (args.variant, t, s),
            "value": clips_s, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            # each rank's own wall clock over the K steps (the barriers around the region equalise them unless a rank is slow
            # OUTSIDE the collectives): a scaling loss with min ~ max is exchange / launch overhead, min << max a slow rank
            "ms_per_step_ranks": {"min": 1e3 * min(per_rank) / args.steps, "max": 1e3 * max(per_rank) / args.steps,
                                  "all": [round(1e3 * v / args.steps, 4) for v in per_rank]},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"X3D-{args.variant} train step (fwd + bwd + Nesterov SGD), clips {t}x{s}x{s}x3, "
                                   f"{args.dtype} activation storage / fp32 accumulation (matrix-core operands in {args.dtype}), random-init weights",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "loss": loss,
            "collectives": coll,
            "roofline": None if dom is None else {
                "bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": dom["achieved_GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                "launches": dom["launches"], "avg_us": dom["avg_us"],
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"]},
            "kernels": kernels,
            "mfma_util": mfma,
            "roofline_model": {"bound": "hbm", "achieved": clips_s / world * step_bytes_per_clip / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": clips_s / world * step_bytes_per_clip / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_clip": step_bytes_per_clip},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.variant, args.cpu_seconds)
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
