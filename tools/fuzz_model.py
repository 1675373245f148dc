"""Whole-model parity on odd configurations (GPU box): tests/test_model_gpu.py's training-step and inference comparisons
against the CPU oracle on clips whose planes are ragged at every stage (91 / 46 / 23 / 12 / 6, 13 frames, 3 / 5 frames).

    python tools/fuzz_model.py
"""
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_model_gpu as M  # noqa: E402

gpu = torch.device("cuda:0")
fails = 0
for case in [("S", 1, 13, 91), ("XS", 2, 3, 91), ("M", 1, 5, 78), ("S", 2, 13, 46), ("XS", 3, 4, 50)]:
    try:
        M.test_train_step_fp32(gpu, *case)
        print("ok fp32 train", case, flush=True)
    except Exception:
        fails += 1
        print("FAIL fp32 train", case)
        traceback.print_exc(limit=3)
for case in [("S", 1, 13, 91), ("XS", 2, 3, 91), ("S", 2, 13, 46), ("M", 1, 5, 78)]:
    for dt in (torch.bfloat16, torch.float16):
        try:
            M.test_train_step_half_block_by_block(gpu, *case, dt)
            print("ok half train", case, dt, flush=True)
        except Exception:
            fails += 1
            print("FAIL half train", case, dt)
            traceback.print_exc(limit=3)
for case in [("S", 2, 3, 13, 91, torch.float16), ("S", 2, 3, 13, 91, torch.float32), ("XS", 3, 1, 4, 91, torch.bfloat16)]:
    try:
        M.test_forward_inference(gpu, *case)
        print("ok inference", case, flush=True)
    except Exception:
        fails += 1
        print("FAIL inference", case)
        traceback.print_exc(limit=3)
print("failures:", fails)
sys.exit(1 if fails else 0)
