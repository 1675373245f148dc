"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

MI355X_MICROARCH.md (HBM): the two counters do not fit one pass; both are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   bytes per launch.
Usage: python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/>  -> JSON on stdout:
    {kernel name: {launches, fetch_KiB_avg, write_KiB_avg, traffic_bytes_per_launch}}
"""
import csv
import glob
import json
import os
import re
import sys


_DW = re.compile(r"_Z\d+(dw3d_\w+?_kernel)I(DF16b|DF16_|f)((?:L[ib]n?\d+E)+)E")
_TYPES = {"DF16b": "bf16", "DF16_": "f16", "f": "float"}


def canonical(name):
    """mangled depthwise instantiation -> the name x3d_dw3d_kernel_name() / bench.py use (rocprofv3 -M keeps
    names mangled; its demangler garbles the __bf16 template argument)."""
    m = _DW.match(name)
    if not m:
        return name
    args = [x.replace("n", "-") for x in re.findall(r"L[ib](n?\d+)E", m.group(3))]   # (Lin8E = -8)
    return "%s<%s, %s>" % (m.group(1), _TYPES[m.group(2)], ", ".join(args))


def collect(d, counter):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = canonical(row["Kernel_Name"])
                a = agg.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return agg


def main():
    root = sys.argv[1]
    fetch = collect(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = collect(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fn, fs = fetch.get(k, [0, 0.0])
        wn, ws = write.get(k, [0, 0.0])
        fa = fs / fn if fn else 0.0
        wa = ws / wn if wn else 0.0
        out[k] = dict(launches=max(fn, wn), fetch_KiB_avg=fa, write_KiB_avg=wa,
                      traffic_bytes_per_launch=(2.0 * fa + wa) * 1024.0)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
