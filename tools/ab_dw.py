"""A/B of the depthwise kernels on the X3D-M layer shapes (GPU box): HIP-event time per launch and a CRC of
every output, so two runs with different experiment switches (e.g. X3D_DW_PD=1 = one-plane-ahead kernels
only) can be compared for speed AND bit-identity.

    python tools/ab_dw.py out.json [batch]
    X3D_DW_PD=1 python tools/ab_dw.py base.json [batch]
    python tools/ab_dw.py --compare base.json out.json
"""
import json
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # C, T, H, W, stride  (X3D-M, plus L's odd 39 -> 20)
    (54, 16, 112, 112, 2), (54, 16, 56, 56, 1), (108, 16, 56, 56, 2), (108, 16, 28, 28, 1), (216, 16, 28, 28, 2),
    (216, 16, 14, 14, 1), (432, 16, 14, 14, 2), (432, 16, 7, 7, 1), (216, 16, 39, 39, 2), (432, 13, 10, 10, 1),
]


def crc(t):
    return zlib.crc32(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    if sys.argv[1] == "--compare":
        a, b = (json.load(open(p)) for p in sys.argv[2:4])
        ok = True
        for k in a:
            same = a[k]["crc"] == b[k]["crc"]
            ok &= same
            print(f"{k:34s} {a[k]['us']:8.1f} -> {b[k]['us']:8.1f} us  {a[k]['us'] / b[k]['us']:5.2f}x  "
                  f"{b[k]['kernel']:44s} {'bit-identical' if same else 'DIFFERENT'}")
        sys.exit(0 if ok else 1)
    from x3d_tf_amd import hip, ops
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    lib = hip.load()
    out = {}
    only = os.environ.get("AB_ONLY")          # e.g. AB_ONLY="216,16,14,14,1": one shape, bf16 only
    shapes = [tuple(int(v) for v in only.split(","))] if only else SHAPES
    for dtype in ((torch.bfloat16,) if only else (torch.bfloat16, torch.float32)):
        for c, t, h, w, s in shapes:
            nn = n if dtype == torch.bfloat16 else max(1, n // 8)
            g = torch.Generator().manual_seed(c * 7 + h)
            ho, wo = -(-h // s), -(-w // s)
            x = torch.randn((nn, c, t, h, w), generator=g).to(dtype).to(dev)
            dv = torch.randn((nn, c, t, ho, wo), generator=g).to(dtype).to(dev)
            braw = torch.randn((nn, c, t, ho, wo), generator=g).to(dtype).to(dev)
            wt = (torch.randn((c, 27), generator=g) * 0.3).to(dev)
            ss = torch.stack([1 + 0.3 * torch.randn(c, generator=g), 0.3 * torch.randn(c, generator=g)], 1).to(dev)
            coef = (torch.randn((nn, c, 4), generator=g) * 0.5).to(dev)
            tag = f"{'bf16' if dtype == torch.bfloat16 else 'fp32'} C{c} {t}x{h}x{w} s{s}"
            # forward
            y = torch.empty((nn, c, t, ho, wo), dtype=dtype, device=dev)
            stats = ops.stats_buffer(c, dev)   # replicated accumulator (include/x3d_hip.h)
            pool = torch.zeros((nn, c), dtype=torch.float64, device=dev)
            fa = hip.Dw3dFwdArgs(hip.ptr(x), hip.ptr(wt), hip.ptr(y), hip.ptr(ss), 1, hip.ptr(stats), hip.ptr(pool),
                                 nn, c, t, h, w, s, hip.dtype_code(dtype))
            ops.dw3d_fwd(x, wt, s, y=y, in_ss=ss, in_act=1, stats=stats, pool=pool)
            torch.cuda.synchronize()
            k = crc(y)   # stats / pool: fp64 atomics, order-dependent in the last bit
            us = timed(lambda: ops.dw3d_fwd(x, wt, s, y=y, in_ss=ss, in_act=1, stats=stats, pool=pool))
            out["fwd " + tag] = {"us": us, "crc": k, "kernel": hip.dw3d_kernel_name(fa)}
            # backward
            ga = torch.empty_like(x)
            a_sums = torch.zeros((c, 2), dtype=torch.float64, device=dev)
            dw = torch.zeros((c, 27), dtype=torch.float32, device=dev)
            ba = hip.Dw3dBwdArgs(hip.ptr(dv), hip.ptr(braw), hip.ptr(coef), hip.ptr(x), hip.ptr(ss), hip.ptr(wt), hip.ptr(ga),
                                 hip.ptr(a_sums), hip.ptr(dw), nn, c, t, h, w, s, hip.dtype_code(dtype))
            ops.dw3d_bwd(dv, braw, coef, x, ss, wt, ga, a_sums, dw, s)
            torch.cuda.synchronize()
            k = crc(ga)   # a_sums / dw: atomics, order-dependent in the last bit
            us = timed(lambda: ops.dw3d_bwd(dv, braw, coef, x, ss, wt, ga, a_sums, dw, s))
            out["bwd " + tag] = {"us": us, "crc": k, "kernel": hip.dw3d_kernel_name(ba)}
            del x, dv, braw, y, ga
            _ = lib
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:34s} {v['us']:8.1f} us  {v['kernel']}")


if __name__ == "__main__":
    main()
