"""(needs a library built with X3D_EXPERIMENTS=1 -- the product build reads its A/B switches once)
In-process A/B of the matrix-core depthwise kernels (dw_mx.hip, X3D_DW_MX) against the vector kernels on the 14x14
stride-1 layer of X3D-M stage 4: time per launch (variants alternating) and the difference of the outputs.

    python tools/ab_mx.py [C,T,H,W ...]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(216, 16, 14, 14), (108, 16, 28, 28), (54, 16, 56, 56)]


def timed(fn, reps=30):
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
    from x3d_tf_amd import hip, ops
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or SHAPES
    dev = torch.device("cuda:0")
    hip.load()
    n = 64
    for dtype in (torch.bfloat16,):
        for c, t, h, w in shapes:
            g = torch.Generator().manual_seed(c * 7 + h)
            x = torch.randn((n, c, t, h, w), generator=g).to(dtype).to(dev)
            dv = torch.randn((n, c, t, h, w), generator=g).to(dtype).to(dev)
            braw = torch.randn((n, c, t, h, w), generator=g).to(dtype).to(dev)
            wt = (torch.randn((c, 27), generator=g) * 0.3).to(dev)
            ss = torch.stack([1 + 0.3 * torch.randn(c, generator=g), 0.3 * torch.randn(c, generator=g)], 1).to(dev)
            coef = (torch.randn((n, c, 4), generator=g) * 0.5).to(dev)
            y = torch.empty_like(x)
            ga = torch.empty_like(x)
            res = {}
            for rnd in range(5):
                for mode in ("0", "1"):
                    os.environ["X3D_DW_MX"] = mode; os.environ["X3D_DW_MXW"] = mode
                    stats = ops.stats_buffer(c, dev)
                    pool = torch.zeros((n, c), dtype=torch.float64, device=dev)
                    f = lambda: ops.dw3d_fwd(x, wt, 1, y=y, in_ss=ss, in_act=1, stats=stats, pool=pool)
                    a_sums = torch.zeros((c, 2), dtype=torch.float64, device=dev)
                    dw = torch.zeros((c, 27), dtype=torch.float32, device=dev)
                    b = lambda: ops.dw3d_bwd(dv, braw, coef, x, ss, wt, ga, a_sums, dw, 1)
                    r = res.setdefault(mode, {"fwd": [], "bwd": []})
                    r["fwd"].append(timed(f))
                    r["bwd"].append(timed(b))
                    if rnd == 0:
                        stats.zero_(); pool.zero_(); a_sums.zero_(); dw.zero_()
                        f(); b()
                        torch.cuda.synchronize()
                        r["y"], r["ga"], r["dw"], r["pool"], r["as"] = y.float().clone(), ga.float().clone(), dw.clone(), pool.clone(), a_sums.clone()
                        fa = hip.Dw3dFwdArgs(hip.ptr(x), hip.ptr(wt), hip.ptr(y), hip.ptr(ss), 1, hip.ptr(stats), hip.ptr(pool),
                                             n, c, t, h, w, 1, hip.dtype_code(dtype))
                        ba = hip.Dw3dBwdArgs(hip.ptr(dv), hip.ptr(braw), hip.ptr(coef), hip.ptr(x), hip.ptr(ss), hip.ptr(wt), hip.ptr(ga),
                                             hip.ptr(a_sums), hip.ptr(dw), n, c, t, h, w, 1, hip.dtype_code(dtype))
                        r["kf"], r["kb"] = hip.dw3d_kernel_name(fa), hip.dw3d_kernel_name(ba)
            for k in ("fwd", "bwd"):
                print(f"{dtype} C{c} {t}x{h}x{w} {k}: vector {statistics.median(res['0'][k]):7.1f} us   matrix {statistics.median(res['1'][k]):7.1f} us"
                      f"   {res['0']['kf' if k == 'fwd' else 'kb']} | {res['1']['kf' if k == 'fwd' else 'kb']}", flush=True)
            for k in ("y", "ga", "dw", "pool", "as"):
                a0, a1 = res["0"][k].double(), res["1"][k].double()
                print(f"   {k:5s} max |diff| {float((a0 - a1).abs().max()):.3e}  of max |ref| {float(a0.abs().max()):.3e}   rel-L2 {float((a0 - a1).norm() / a0.norm()):.2e}")


if __name__ == "__main__":
    main()
