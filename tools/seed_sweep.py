"""Seed sweep of the teacher-forced 16-bit model test on the headline's planes (X3D-M 1 x 4 x 224^2), with the
recomputed-output backward on (the product) and off: the evidence behind the per-tensor limits of
tests/test_model_gpu.py::test_train_step_half_block_by_block.  One process, every case in turn.

    python tools/seed_sweep.py [first_seed last_seed] > profiles/rNN_seed_sweep.log      (on the GPU box)
"""
import io
import os
import sys
from contextlib import redirect_stdout

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_model_gpu as T  # noqa: E402

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 7)
gpu = torch.device("cuda:0")
for dtype in (torch.bfloat16, torch.float16):
    for rc in ("1", "0"):
        for seed in range(lo, hi + 1):
            os.environ["X3D_TEST_SEED"], os.environ["X3D_TEST_RC"] = str(seed), rc
            buf, verdict = io.StringIO(), "pass"
            try:
                with redirect_stdout(buf):
                    T.test_train_step_half_block_by_block(gpu, "M", 1, 4, 224, dtype)
            except AssertionError as e:
                verdict = "FAIL " + str(e)[:300]
            line = [l for l in buf.getvalue().splitlines() if l.startswith("teacher-forced worst")]
            print(f"{str(dtype):16s} rc={rc} seed={seed}: {verdict} | {line[-1] if line else ''}", flush=True)
