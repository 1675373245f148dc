"""Which buffers does a backward launch list WRITE?  Checksums of every tensor a training plan references -- forward
activations, BatchNorm coefficient tables, SE gates, parameters, the fp64 accumulators -- before and after a replay of
the backward list (the product's, and the one recorded with other plan options).  A backward list may change its own scratch,
the gradients and the accumulators it owns; a forward tensor or a parameter that changes is an out-of-bounds write.

    python tools/debug_bwd_writes.py [variant N T S dtype] [option=0/1 ...]      (on the GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x3d_tf_amd as x  # noqa: E402
from x3d_tf_amd import hip  # noqa: E402
from x3d_tf_amd.model import X3D  # noqa: E402
from x3d_tf_amd.params import init_params, randomize_bn_  # noqa: E402
from tests.util import record_alternate_backward  # noqa: E402


def named_forward_tensors(m, pl):
    out = {"flat_params": m.flat_params, "x": pl._x_keepalive, "s_raw": pl.s_raw, "t_raw": pl.t_raw, "y0": pl.y0,
           "c5_raw": pl.c5_raw, "pooled": pl.pooled, "h1": pl.h1, "logits": pl.logits, "probs": pl.probs, "dlogits": pl.dlogits,
           "labels": pl.labels, "loss_rows": pl.loss_rows}
    if pl.drop_mask is not None:
        out["drop_mask"] = pl.drop_mask
    if m._panel_table is not None:
        out["panel_buf"] = m._panel_buf
    for nm, bn in (("bn1", pl.bn1), ("bn5", pl.bn5)):
        out[nm + ".ss"], out[nm + ".mi"] = bn.ss, bn.mi
    zv = pl._zero_views
    for i, B in enumerate(pl.blocks):
        p = f"block{i}."
        if B.pool is not None:
            out[p + "pool(acc)"] = zv[B.pool]
        for k in ("bn_a", "bn_b", "bn_c", "bn_r"):
            bn = getattr(B, k, None)
            if bn is not None and bn.stats is not None:
                out[p + k + ".stats(acc)"] = zv[bn.stats]
        for k in ("a_raw", "b_raw", "c_raw", "y", "r_raw", "gate", "hidden"):
            v = getattr(B, k, None)
            if isinstance(v, torch.Tensor):
                out[p + k] = v
        for k in ("bn_a", "bn_b", "bn_c", "bn_r"):
            bn = getattr(B, k, None)
            if bn is not None:
                out[p + k + ".ss"], out[p + k + ".mi"] = bn.ss, bn.mi
    return out


def digest(t):
    v = t.detach().contiguous().reshape(-1).view(torch.uint8)
    pad = (-v.numel()) % 8
    if pad:
        v = torch.cat([v, torch.zeros(pad, dtype=torch.uint8, device=v.device)])
    w = v.view(torch.int64)
    idx = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
    return int((w * (2 * idx + 1)).sum().item()), int(w.sum().item())


def main():
    a = sys.argv[1:]
    pos = [s for s in a if "=" not in s]
    opts = {s.split("=")[0]: s.split("=")[1] == "1" for s in a if "=" in s}
    variant, n, t, s = (pos + ["M", "1", "4", "224"])[:4]
    n, t, s = int(n), int(t), int(s)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[(pos + [None] * 5)[4] or "bf16"]
    dev = torch.device("cuda:0")
    cfg = x.get_config(variant)
    arch = x.build_arch(cfg)
    m = X3D(cfg, dtype=dtype, device=dev)
    m.load_state_dict(randomize_bn_(init_params(arch, seed=3), seed=4))
    torch.manual_seed(2)
    xin = torch.randn(n, t, s, s, 3).to(dtype).to(dev)
    labels = torch.randint(0, arch.num_classes, (n,)).to(dev)
    pl = m.forward_backward(xin, labels)
    torch.cuda.synchronize()
    alt = record_alternate_backward(m, pl, xin, **(opts or {"pw_bwd_rc": False}))
    m._pack_panels()
    pl.zero_buf.zero_()
    pl.run(pl.fwd, 0, pl.grad_scale_slot)
    hip.call("x3d_softmax_xent", pl.logits.data_ptr(), pl.labels.data_ptr(), pl.probs.data_ptr(), pl.loss_rows.data_ptr(),
             pl.dlogits.data_ptr(), 1.0 / n, n, arch.num_classes)
    torch.cuda.synchronize()
    snap = pl.zero_buf.clone()
    names = named_forward_tensors(m, pl)
    # the forward part of the accumulator buffer: everything allocated before the backward pass was recorded the first time
    base = {k: digest(v) for k, v in names.items()}
    print(f"{len(names)} forward-side tensors tracked; accumulator buffer {pl.zero_buf.numel()} doubles")
    for tag, runner in (("product list", lambda: pl.run(pl.bwd)), (f"alternate list {opts or {'pw_bwd_rc': False}}", alt.run),
                        ("product list again", lambda: pl.run(pl.bwd))):
        pl.zero_buf.copy_(snap)
        m.flat_grads.zero_()
        runner()
        torch.cuda.synchronize()
        changed = [k for k, v in names.items() if digest(v) != base[k]]
        print(f"{tag}: forward-side tensors changed: {changed if changed else 'none'}")
        for k in changed:
            cur = names[k]
            print("   ", k, tuple(cur.shape), cur.dtype, "ptr", hex(cur.data_ptr()), "bytes", cur.numel() * cur.element_size())
        base = {k: digest(v) for k, v in names.items()}
    if "--bisect" in sys.argv:
        # launch by launch: the first launch of each list after which `snap` (a bystander allocation) or the forward part of the
        # accumulator buffer differs
        from x3d_tf_amd.dispatch import describe_struct
        fwd_len = min(v.data_ptr() for k, v in names.items() if k.endswith("(acc)")), max(v.data_ptr() + v.numel() * 8 for k, v in names.items() if k.endswith("(acc)"))
        lo = (fwd_len[0] - pl.zero_buf.data_ptr()) // 8
        hi = (fwd_len[1] - pl.zero_buf.data_ptr()) // 8
        print(f"forward accumulators: doubles [{lo}, {hi}) of {pl.zero_buf.numel()}; snap at {hex(snap.data_ptr())}, zero_buf at {hex(pl.zero_buf.data_ptr())}, extra at {hex(alt.extra.data_ptr())}")
        snap2 = snap.clone()
        for tag, lst in (("product", pl.bwd), ("alternate", alt.lst), ("product again", pl.bwd)):
            pl.zero_buf.copy_(snap2)
            snap.copy_(snap2)
            m.flat_grads.zero_()
            alt.extra.zero_()
            torch.cuda.synchronize()
            d_snap, d_fwd = digest(snap), digest(pl.zero_buf[lo:hi])
            stream = torch.cuda.current_stream().cuda_stream
            for i, (name, fn, args) in enumerate(lst):
                rc = fn(*args, stream)
                assert rc == 0, (name, rc)
                torch.cuda.synchronize()
                a, b = digest(snap), digest(pl.zero_buf[lo:hi])
                if a != d_snap or b != d_fwd:
                    st = pl.structs.get((id(lst), i))
                    print(f"{tag} list: launch {i} {name} {describe_struct(st) if st is not None and hasattr(st, 'N') else ''} changed "
                          f"{'snap ' if a != d_snap else ''}{'forward accumulators' if b != d_fwd else ''}")
                    if b != d_fwd:
                        diff = (pl.zero_buf[lo:hi] != snap2[lo:hi]).nonzero().flatten()
                        print(f"    {diff.numel()} doubles differ, first {int(diff[0]) + lo} last {int(diff[-1]) + lo}; values now {pl.zero_buf[lo:hi][diff[:4]].tolist()} were {snap2[lo:hi][diff[:4]].tolist()}")
                    d_snap, d_fwd = a, b
            print(f"{tag} list: done")
    order = sorted(names.items(), key=lambda kv: kv[1].data_ptr())
    if "--map" in sys.argv:
        for k, v in order:
            print(f"  {hex(v.data_ptr())} .. {hex(v.data_ptr() + v.numel() * v.element_size())}  {k}")


if __name__ == "__main__":
    main()
