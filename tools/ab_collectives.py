"""Where does the per-step cost of the data-parallel collectives come from BEFORE a byte crosses xGMI?  One rank, RCCL communicator
of world size 1 (X3D_DIST_REHEARSE=1: every collective runs, the sums are identities), one process, one box: the same Trainer
stepped in turn with
  none          no hook, no collective (the single-GPU step)
  hooks_only    the backward list cut at the six stage marks and the hook called, nothing launched
  stats_only    only the moving-statistics all-reduce (launched behind the forward pass)
  one_bucket    one all-reduce of the whole gradient buffer behind the last backward kernel (fully exposed)
  six_buckets   the product's six buckets from the backward marks, no moving statistics
  product       six buckets + moving statistics
  side_wait / side_kernel   no torch.distributed at all: a second stream that waits for the forward pass (and runs one tiny kernel),
                waited for in front of the optimizer -- the bare cost of a second hardware queue in the step
  record_only / fork_only / join_only   its parts: an event recorded behind the forward pass that nobody waits for; the side stream
                waiting for the forward pass, never joined; the compute stream waiting for the idle side stream in front of the optimizer
each timed over `steps` steps, rounds alternating.  RCCL's channel count is fixed when the communicator is made: run the tool once
per NCCL_MAX_NCHANNELS setting.

    X3D_DIST_REHEARSE=1 [NCCL_MAX_NCHANNELS=k] python tools/ab_collectives.py [steps=20] [rounds=3]      (on the GPU box)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("X3D_DIST_REHEARSE", "1")
import x3d_tf_amd as x3d  # noqa: E402
from x3d_tf_amd import dist as xdist  # noqa: E402
from x3d_tf_amd.model import X3D  # noqa: E402
from x3d_tf_amd.train import Trainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    xdist.init_process_group()
    dev = torch.device("cuda:0")
    cfg = x3d.get_config("M")
    model = X3D(cfg, dtype=torch.bfloat16, device=dev, seed=0)
    tr = Trainer(model, cfg)
    assert tr.collectives, "no process group: run with X3D_DIST_REHEARSE=1"
    g = torch.Generator(device=dev).manual_seed(1000)
    clips = torch.randn((64, 16, 224, 224, 3), generator=g, device=dev).to(torch.bfloat16)
    labels = torch.randint(0, model.num_classes, (64,), generator=g, device=dev)
    lr = cfg.TRAIN.WARMUP_LR
    six, at6 = list(tr.reducer.buckets), dict(tr._launch_at)
    whole = model.flat_grads[:model.n_trainable_flat]

    def setup(arm):
        state["arm"] = arm
        tr.collectives = arm != "none"
        tr.sync_moving_stats = arm in ("stats_only", "product")
        if arm in ("six_buckets", "product"):
            tr.reducer.buckets, tr._launch_at = six, at6
        elif arm == "one_bucket":
            tr.reducer.buckets, tr._launch_at = [whole], {-1: 0}
        else:
            tr.reducer.buckets, tr._launch_at = six, {}
        tr.reducer.active = tr.collectives

    # two arms without torch.distributed: the same stream pattern by hand -- a side stream that waits for the forward pass, runs
    # one tiny kernel (side_kernel) or nothing (side_wait), and is waited for in front of the optimizer
    side = torch.cuda.Stream()
    tiny = torch.zeros(1024, device=dev)
    real_hook, real_finish = tr._on_stage_done, tr.reducer.finish
    state = {"arm": "none"}

    ev = torch.cuda.Event()
    HAND = ("side_kernel", "side_wait", "record_only", "fork_only", "join_only")

    def hook(stage):
        if state["arm"] in HAND:
            if stage == "fwd":
                if state["arm"] == "record_only":           # an event recorded on the compute stream, nobody waits for it
                    ev.record()
                elif state["arm"] != "join_only":           # fork: the side stream waits for the forward pass
                    side.wait_stream(torch.cuda.current_stream())
                if state["arm"] == "side_kernel":
                    with torch.cuda.stream(side):
                        tiny.add_(1.0)
            return
        real_hook(stage)

    def finish():
        if state["arm"] in HAND:
            if state["arm"] in ("side_kernel", "side_wait", "join_only"):   # join: the compute stream waits for the (idle) side stream
                torch.cuda.current_stream().wait_stream(side)
            return
        real_finish()

    tr._on_stage_done, tr.reducer.finish = hook, finish
    arms = ["none", "hooks_only", "record_only", "fork_only", "join_only", "side_wait", "side_kernel", "stats_only", "one_bucket", "six_buckets", "product"]
    for a in arms:          # warm every arm (communicator, plan, autotuned RCCL kernels)
        setup(a)
        for _ in range(3):
            tr.step(clips, labels, lr)
    torch.cuda.synchronize()
    res = {a: [] for a in arms}
    for r in range(rounds):
        for a in arms:
            setup(a)
            tr.step(clips, labels, lr)
            torch.cuda.synchronize()
            tr.reducer.exposed_ms()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(clips, labels, lr)
            host = (time.perf_counter() - t0) * 1e3 / steps       # host time to ENQUEUE a step (the device runs behind)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / steps
            ex = tr.reducer.exposed_ms()
            res[a].append(ms)
            print(f"round {r} {a:12s} {ms:8.3f} ms/step   host enqueue {host:7.3f} ms/step   exposed {ex if ex is None else round(ex, 3)}", flush=True)
    base = min(res["none"])
    print(f"# NCCL_MAX_NCHANNELS={os.environ.get('NCCL_MAX_NCHANNELS', 'default')}  best of {rounds} rounds, {steps} steps each")
    for a in arms:
        print(f"# {a:12s} {min(res[a]):8.3f} ms/step   {min(res[a]) - base:+.3f} ms vs none")


if __name__ == "__main__":
    main()
