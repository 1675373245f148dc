"""Static check for the in-order-retirement trap: `s_waitcnt vmcnt(0)` INSIDE a loop of a kernel means some
use in the loop waits for every global access in flight -- prefetches issued for later iterations included -- typically
because a global load was issued next to its use (behind the prefetches) or because a conditional access keeps the compiler
from counting.  Lists, per kernel of an assembly listing (hipcc -S --cuda-device-only), the loop blocks that contain one.

    python tools/scan_waitcnt.py file.s [name filter]
"""
import re
import subprocess
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    fn, in_loop, hits, loads = None, False, {}, {}
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn, in_loop = m.group(1), False
            continue
        if fn is None:
            continue
        if line.startswith(".LBB") or line.startswith("; %bb.") or line.startswith(".Lfunc_end"):
            in_loop = ("in Loop" in line) or ("Loop Header" in line)
            if line.startswith(".Lfunc_end"):
                fn = None
            continue
        if in_loop and "s_waitcnt" in line and "vmcnt(0)" in line:
            hits[fn] = hits.get(fn, 0) + 1
        if in_loop and re.search(r"\b(global|buffer)_load", line):
            loads[fn] = loads.get(fn, 0) + 1
    names = list(hits)
    if not names:
        print("no vmcnt(0) inside loops")
        return
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    for n, d in sorted(zip(names, dem), key=lambda t: -hits[t[0]]):
        d = d.replace("__hip_bfloat16", "bf16").replace("(PwGemmArgs)", "").replace("void ", "")
        if filt in d:
            print(f"{hits[n]:3d} x vmcnt(0) in loops ({loads.get(n, 0)} loads in loops)  {d[:150]}")


if __name__ == "__main__":
    main()
