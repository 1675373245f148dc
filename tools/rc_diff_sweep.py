"""Seed sweep of the recomputed-output differential test (tests/test_model_gpu.py::
test_recomputed_output_backward_against_the_stored_output_backward): both shapes, both 16-bit storage types, seeds lo .. hi.
The limits RC_DIFF_LIMITS are set from the spread this prints (the worst tensors are cancelling sums whose error is a
heavy-tailed sample of the operand rounding, different for every forward state).

    python tools/rc_diff_sweep.py [first_seed last_seed] > profiles/rNN_rc_diff_sweep.log      (on the GPU box)
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
    for name, n, t, s in (("M", 1, 4, 224), ("L", 1, 2, 312)):
        for seed in range(lo, hi + 1):
            os.environ["X3D_TEST_SEED"] = str(seed)
            buf, verdict = io.StringIO(), "pass"
            try:
                with redirect_stdout(buf):
                    T.test_recomputed_output_backward_against_the_stored_output_backward(gpu, name, n, t, s, dtype)
            except AssertionError as e:
                verdict = "FAIL " + str(e)[:400]
            for line in buf.getvalue().splitlines():
                if line.startswith("rc differential"):
                    print(f"seed={seed} {line[:330]}", flush=True)
            print(f"{str(dtype):16s} {name} seed={seed}: {verdict}", flush=True)
