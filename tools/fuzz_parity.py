"""Randomised parity sweep on the GPU box: the kernel-level parity tests of tests/test_kernels_gpu.py (depthwise forward /
fused backward, pointwise forward / data gradient / weight gradient) on RANDOM shapes -- odd widths, ragged rows, point
counts that are not a multiple of 8, strips cut by row ends -- in all three storage types.  A failure prints the shape.

    python tools/fuzz_parity.py [cases] [seed]
"""
import os
import random
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_kernels_gpu as K  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    gpu = torch.device("cuda:0")
    fails = 0
    widths = [3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 17, 19, 20, 21, 23, 26, 28, 31, 37, 39, 40, 45, 46, 53, 56, 57, 78, 91]
    chans = [24, 48, 54, 96, 108, 192, 216, 432, 40, 72, 200]
    for i in range(cases):
        dt = rng.choice(DTYPES)
        # depthwise
        shp = (rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 5]), rng.choice([1, 2, 3, 5, 8]), rng.choice(widths), rng.choice(widths),
               rng.choice([1, 2]))
        for fn in (K.test_dw3d_fwd, K.test_dw3d_bwd):
            try:
                fn(gpu, dt, shp)
            except Exception:
                fails += 1
                print("FAIL", fn.__name__, dt, shp)
                traceback.print_exc(limit=2)
        # pointwise: T*H*W any
        t, h, w = rng.choice([1, 2, 3, 5, 13]), rng.choice([3, 5, 7, 8, 10, 12, 14]), rng.choice([3, 5, 7, 8, 10, 12, 14])
        cin, cout = rng.choice(chans), rng.choice(chans)
        n = rng.choice([1, 2])
        try:
            K.test_pw_fwd(gpu, dt, (n, cin, cout, t, h, w, 1, rng.choice([None, "swish", "relu"])), rng.choice([False, True]) and dt != torch.float32)
        except Exception:
            fails += 1
            print("FAIL pw_fwd", dt, (n, cin, cout, t, h, w))
            traceback.print_exc(limit=2)
        try:
            epi = rng.choice(["store", "add", "add_strided", "swish_bwd"])
            K.test_pw_dgrad(gpu, dt, (n, cin, cout, t, h, w), epi, rng.choice([False, True]) and dt != torch.float32)
        except Exception:
            fails += 1
            print("FAIL pw_dgrad", dt, (n, cin, cout, t, h, w), epi)
            traceback.print_exc(limit=2)
        try:
            K.test_pw_wgrad(gpu, dt, (n, cin, cout, t, h, w, 1, rng.choice([None, "swish"])))
        except Exception:
            fails += 1
            print("FAIL pw_wgrad", dt, (n, cin, cout, t, h, w))
            traceback.print_exc(limit=2)
        # strided shortcut (stride 2, no prologue): odd and even input widths, P a multiple of 8 or not
        ts, hs, ws = rng.choice([1, 2, 4, 8]), rng.choice([5, 8, 9, 13, 16, 20, 39]), rng.choice([7, 8, 11, 13, 16, 23, 27, 39, 40, 46])
        try:
            K.test_pw_fwd(gpu, dt, (n, rng.choice([24, 32, 48]), rng.choice([24, 48, 96]), ts, hs, ws, 2, None), False)
        except Exception:
            fails += 1
            print("FAIL pw_fwd strided", dt, (n, ts, hs, ws))
            traceback.print_exc(limit=2)
        try:
            K.test_pw_wgrad(gpu, dt, (n, rng.choice([24, 32, 48]), rng.choice([24, 48, 96]), ts, hs, ws, 2, None))
        except Exception:
            fails += 1
            print("FAIL pw_wgrad strided", dt, (n, ts, hs, ws))
            traceback.print_exc(limit=2)
        if (i + 1) % 10 == 0:
            print(f"{i + 1} cases, {fails} failures", flush=True)
    print(f"done: {cases} cases, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
