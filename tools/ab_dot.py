"""(needs a library built with X3D_EXPERIMENTS=1 -- the product build reads its A/B switches once)
In-process A/B of the depthwise backward with the tap loops on two-element dot products (X3D_DW_DOT) against the
scalar-FMA form: the variants alternate inside one process on the same tensors (box-to-box and run-to-run spread of
single runs is +-5 %), medians over rounds.

    python tools/ab_dot.py [C,T,H,W,stride ...]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(216, 16, 14, 14, 1), (432, 16, 7, 7, 1), (108, 16, 28, 28, 1), (54, 16, 56, 56, 1)]
MODES = [("fma", {"X3D_DW_DOT": "0"}), ("dot dW", {"X3D_DW_DOT": "1", "X3D_DW_DOTMASK": "1"}),
         ("dot dA", {"X3D_DW_DOT": "1", "X3D_DW_DOTMASK": "2"}), ("dot both", {"X3D_DW_DOT": "1", "X3D_DW_DOTMASK": "3"})]


def main():
    from x3d_tf_amd import hip, ops
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or SHAPES
    dev = torch.device("cuda:0")
    hip.load()
    n, dtype = 64, torch.bfloat16
    for c, t, h, w, s in shapes:
        g = torch.Generator().manual_seed(c * 7 + h)
        ho, wo = -(-h // s), -(-w // s)
        x = torch.randn((n, c, t, h, w), generator=g).to(dtype).to(dev)
        dv = torch.randn((n, c, t, ho, wo), generator=g).to(dtype).to(dev)
        braw = torch.randn((n, c, t, ho, wo), generator=g).to(dtype).to(dev)
        wt = (torch.randn((c, 27), generator=g) * 0.3).to(dev)
        ss = torch.stack([1 + 0.3 * torch.randn(c, generator=g), 0.3 * torch.randn(c, generator=g)], 1).to(dev)
        coef = (torch.randn((n, c, 4), generator=g) * 0.5).to(dev)
        ga = torch.empty_like(x)
        a_sums = torch.zeros((c, 2), dtype=torch.float64, device=dev)
        dw = torch.zeros((c, 27), dtype=torch.float32, device=dev)
        ba = hip.Dw3dBwdArgs(hip.ptr(dv), hip.ptr(braw), hip.ptr(coef), hip.ptr(x), hip.ptr(ss), hip.ptr(wt), hip.ptr(ga),
                             hip.ptr(a_sums), hip.ptr(dw), n, c, t, h, w, s, hip.dtype_code(dtype))
        fn = lambda: ops.dw3d_bwd(dv, braw, coef, x, ss, wt, ga, a_sums, dw, s)
        times = {m: [] for m, _ in MODES}
        names = {}
        for rnd in range(7):
            for m, env in MODES:
                os.environ.update(env)
                names[m] = hip.dw3d_kernel_name(ba)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[m].append(e0.elapsed_time(e1) / 30 * 1e3)
        for m, _ in MODES:
            print(f"C{c} {t}x{h}x{w} s{s}  {m:9s} {statistics.median(times[m]):8.1f} us (min {min(times[m]):.1f})  {names[m]}", flush=True)


if __name__ == "__main__":
    main()
