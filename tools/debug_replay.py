"""Debug aid: run one training step, snapshot every stored forward tensor, replay the forward list and report which tensors
differ between the two passes (a backward launch that writes outside its buffers, or a forward that is not a pure function
of its inputs, shows up here).    python tools/debug_replay.py [variant] [n] [t] [s] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x3d_tf_amd as x  # noqa: E402
from x3d_tf_amd.model import X3D  # noqa: E402
from x3d_tf_amd.params import init_params, randomize_bn_  # noqa: E402

dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def snap(pl):
    d = {"y0": pl.y0, "t_raw": pl.t_raw, "c5_raw": pl.c5_raw, "probs": pl.probs}
    for i, B in enumerate(pl.blocks):
        for k in ("a_raw", "b_raw", "c_raw", "y", "r_raw", "gate"):
            v = getattr(B, k, None)
            if v is not None:
                d[f"b{i}.{k}"] = v
        for k in ("bn_a", "bn_b", "bn_c", "bn_r"):
            v = getattr(B, k, None)
            if v is not None:
                d[f"b{i}.{k}.ss"] = v.ss
                if v.stats is not None:
                    d[f"b{i}.{k}.stats"] = pl._zero_views[v.stats]
        if B.pool is not None:
            d[f"b{i}.pool"] = pl._zero_views[B.pool]
    return {k: v.detach().double().cpu().clone() for k, v in d.items()}


KEEP = []


def run(variant, n, t, s, dtype):
    cfg = x.get_config(variant)
    arch = x.build_arch(cfg)
    params = randomize_bn_(init_params(arch, seed=3), seed=4)
    m = X3D(cfg, dtype=dtype, device=dev)
    m.load_state_dict(params)
    torch.manual_seed(2)
    clips = torch.randn(n, t, s, s, 3).to(dtype).float()
    labels = torch.randint(0, arch.num_classes, (n,))
    m.set_dropout_mask((torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float())
    print("====", variant, n, t, s, dtype)
    pl = m.forward_backward(clips.to(dev), labels.to(dev))
    torch.cuda.synchronize()
    s1 = snap(pl)
    panels1 = m._panel_buf.float().cpu().clone() if getattr(m, "_panel_buf", None) is not None else None
    params1 = m.flat_params.cpu().clone()
    pl.zero_buf.zero_()
    pl.run(pl.fwd, 0, pl.grad_scale_slot)
    torch.cuda.synchronize()
    s2 = snap(pl)
    bad = 0
    for k in s1:
        d = (s1[k] - s2[k]).abs().max().item()
        if d > 0:
            bad += 1
            print(f"DIFF {k}: max |first - replay| = {d:.4g} (scale {s1[k].abs().max().item():.4g}), {int((s1[k] != s2[k]).sum())}/{s1[k].numel()} elements")
    print("params changed:", int((params1 != m.flat_params.cpu()).sum()),
          "panel elements changed:", None if panels1 is None else int((panels1 != m._panel_buf.float().cpu()).sum()))
    print("tensors differing between the first pass and the replay:", bad, "of", len(s1))
    KEEP.append((m, pl))      # (as a failed pytest case keeps its locals alive)


args = sys.argv[1:] or ["M", "2", "4", "128", "bf16"]
for i in range(0, len(args), 5):
    run(args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3]), DT[args[i + 4]])
