"""Randomised parity sweep on the GPU box (the generator is tests/fuzz.py; a seeded 200-case slice of it runs inside the
`-m gpu` suite as tests/test_fuzz_gpu.py).  A failure prints the shape.

    python tools/fuzz_parity.py [cases] [seed] [logfile]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fuzz  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    logf = open(sys.argv[3], "w") if len(sys.argv) > 3 else None

    def log(msg):
        print(msg, flush=True)
        if logf:
            logf.write(msg + "\n")
            logf.flush()
    log(f"fuzz_parity: {cases} cases x 12 kernel checks, seed {seed}, {torch.cuda.get_device_name(0)}")
    fails = fuzz.run_cases(torch.device("cuda:0"), cases, seed, log)
    log(f"done: {cases} cases ({12 * cases} kernel checks), {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
