"""Randomised kernel-parity cases: the kernel-level tests of tests/test_kernels_gpu.py (depthwise forward / fused backward,
pointwise forward / data gradient / weight gradient, strided shortcut) on RANDOM shapes -- odd widths, ragged rows, point
counts that are not a multiple of 8, strips cut by row ends, channel counts off the 32-grid -- in all three storage types.
One generator for the in-suite slice (tests/test_fuzz_gpu.py) and the long sweep (tools/fuzz_parity.py)."""
import random

import torch

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
WIDTHS = [3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 17, 19, 20, 21, 23, 26, 28, 31, 37, 39, 40, 45, 46, 53, 56, 57, 78, 91]
CHANS = [24, 48, 54, 96, 108, 192, 216, 432, 40, 72, 200]


def case_calls(rng: random.Random):
    """One fuzz case = 12 kernel-test calls: [(test function name, positional args after `gpu`)]."""
    dt = rng.choice(DTYPES)
    out = []
    shp = (rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 5]), rng.choice([1, 2, 3, 5, 8]), rng.choice(WIDTHS), rng.choice(WIDTHS),
           rng.choice([1, 2]))
    out.append(("test_dw3d_fwd", (dt, shp)))
    out.append(("test_dw3d_bwd", (dt, shp)))
    # the planes the matrix-core depthwise kernels take (dw_mx.hip): 12 / 14 rows and columns, 7 x 7 four to a tile (any sample
    # count), rows of 26 .. 30 elements in H-tiles of 14 rows; T around the prefetch depth and the exit-free loop (T % 4 == 0)
    # (round 4: + ragged rows of 26 columns and more -- windows of 2 / 3 column tiles, W-tiles, odd row lengths: dw_mxg.hip)
    hm, wm = rng.choice([(14, 14), (12, 14), (14, 12), (12, 12), (7, 7), (7, 7), (28, 28), (26, 28), (28, 30), (14, 26), (24, 28),
                         (39, 39), (22, 30), (28, 29), (36, 45), (28, 54), (14, 55), (22, 78), (36, 44), (28, 91), (40, 39), (13, 31)])
    shm = (rng.choice([1, 2, 3, 4, 5, 6]), rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 4, 5, 8, 9]), hm, wm, 1)
    out.append(("test_dw3d_fwd", (dt, shm)))
    out.append(("test_dw3d_bwd", (dt, shm)))
    t, h, w = rng.choice([1, 2, 3, 5, 13]), rng.choice([3, 5, 7, 8, 10, 12, 14]), rng.choice([3, 5, 7, 8, 10, 12, 14])
    cin, cout = rng.choice(CHANS), rng.choice(CHANS)
    n = rng.choice([1, 2])
    half = dt != torch.float32
    out.append(("test_pw_fwd", (dt, (n, cin, cout, t, h, w, 1, rng.choice([None, "swish", "relu"])), rng.choice([False, True]) and half)))
    epi = rng.choice(["store", "add", "add_strided", "swish_bwd"])
    out.append(("test_pw_dgrad", (dt, (n, cin, cout, t, h, w), epi, rng.choice([False, True]) and half)))
    out.append(("test_pw_wgrad", (dt, (n, cin, cout, t, h, w, 1, rng.choice([None, "swish"])))))
    # strided shortcut (stride 2, no prologue): odd and even input widths, P a multiple of 8 or not
    ts, hs, ws = rng.choice([1, 2, 4, 8]), rng.choice([3, 5, 7, 8, 9, 13, 16, 20, 39]), rng.choice([3, 7, 8, 11, 13, 16, 23, 27, 39, 40, 46])
    out.append(("test_pw_fwd", (dt, (n, rng.choice([24, 32, 48]), rng.choice([24, 48, 96]), ts, hs, ws, 2, None), False)))
    out.append(("test_pw_wgrad", (dt, (n, rng.choice([24, 32, 48]), rng.choice([24, 48, 96]), ts, hs, ws, 2, None))))
    # round 5: planes whose rows are whole 16-byte vectors (the stride-2 / stride-1 ring kernels of dw_s2.hip / dw_s1.hip, the
    # de-interleaved stride-2 forward): T around the 6-fold unrolled loop and its drain, partial H-tiles, non-square planes
    shv = (rng.choice([1, 2, 3]), rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 5, 6, 7, 8, 12, 13]),
           rng.choice([8, 14, 20, 27, 28, 33, 40, 45, 56, 64]), rng.choice([16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 120]), rng.choice([1, 2]))
    out.append(("test_dw3d_fwd", (dt, shv)))
    out.append(("test_dw3d_bwd", (dt, shv)))
    # round 6: the fused stem (stem_fused.hip, 16-bit storage only): rows of whole 16-byte vectors, T around the six-fold unrolled
    # time loop and below the kernel length, one to three segments per row, partial last groups, 8 .. 32 channels
    dh = dt if half else rng.choice([torch.bfloat16, torch.float16])
    out.append(("test_stem_fused", (dh, (rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 4, 5, 6, 7, 11, 12, 13, 16]), rng.choice([2, 3, 5, 8, 9, 14, 21]),
                                        rng.choice([8, 16, 24, 40, 56, 72, 104, 128, 136, 160, 264]), rng.choice([8, 16, 24, 24, 32])))))
    return out


def run_cases(gpu, cases: int, seed: int, log=None):
    """Runs `cases` fuzz cases; returns the list of failures as (name, args, message)."""
    import inspect
    import traceback
    from tests import test_kernels_gpu as K
    rng = random.Random(seed)
    fails = []
    for i in range(cases):
        for name, args in case_calls(rng):
            try:
                fn = getattr(K, name)
                extra = {k: v for k, v in (("slab", False), ("jobs", 0)) if k in inspect.signature(fn).parameters}   # (parametrised switches)
                fn(gpu, *args, **extra)
            except Exception as e:   # noqa: BLE001  (a failing case must not stop the sweep)
                fails.append((name, args, f"{type(e).__name__}: {str(e)[:300]}"))
                if log:
                    log(f"FAIL {name} {args}\n{traceback.format_exc(limit=2)}")
        if log and (i + 1) % 10 == 0:
            log(f"{i + 1} cases ({12 * (i + 1)} kernel checks), {len(fails)} failures")
    return fails
