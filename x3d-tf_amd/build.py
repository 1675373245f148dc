"""Builds libx3d_hip.so (hipcc, --offload-arch=gfx950) in-tree.  No JIT cache: the .so lives next to
the sources so it travels with the repository snapshot to the GPU box.

Incremental by CONTENT, not by mtime: every object carries a stamp = SHA-256 over its source, every header and the
compiler flags, the library a stamp over the object stamps.  A checkout that leaves object files newer than edited
sources (or a stale .so travelling with a snapshot) is rebuilt; an untouched tree is not."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libx3d_hip.so")
SOURCES = ["api.cpp", "pw_fwd.hip", "pw_fwd_infer.hip", "pw_fwd_tail.hip", "pw_dgrad.hip", "pw_wgrad.hip", "pw_pack.hip", "pw_bwd_fused.hip", "pw_bwd_wst.hip", "pw_bwd_wsta.hip", "pw_bwd_rc.hip", "dw_fwd.hip", "dw_bwd.hip", "dw_pd.hip", "dw_s1.hip", "dw_s2.hip", "dw_pk.hip", "dw_mx.hip", "dw_mxg.hip", "elem.hip", "stem.hip", "stem_fused.hip",
           "se.hip", "head.hip", "views.hip"]
HEADERS = ["common.h", "pw_gemm.h", "pw_gemm_bf16.h", "pw_gemm_ws.h", "pw_gemm_wst.h", "pw_gemm_f32r.h", "pw_gemm_f32p.h", "pw_wgrad_f32r.h", "pw_wgrad_f32p.h", "pw_wgrad_bf16.h", "dw_common.h", os.path.join("..", "..", "include", "x3d_hip.h")]
assert sorted(h for h in os.listdir(CSRC) if h.endswith(".h")) == sorted(h for h in HEADERS if os.sep not in h), "HEADERS out of date"
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value",
         # hipcc SLP-packs adjacent fp32 mul/add into v_pk_mul + v_pk_add (no FMA, weights no longer SGPR operands):
         # 2.4x the VALU instructions and 2x the VGPRs in the depthwise stencils (measured in the .s)
         "-fno-slp-vectorize"]
if os.environ.get("X3D_EXPERIMENTS") == "1":   # compiles the result-changing timing hooks in (X3D_PW_WG_NOFLUSH, X3D_DW_PK_NOLOAD)
    FLAGS.append("-DX3D_EXPERIMENTS")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.relpath(p, HERE).encode() + b"\0" + f.read())   # relative: a snapshot copied elsewhere is not rebuilt
    return h.hexdigest()


def _stamp_ok(target, want):
    try:
        return os.path.exists(target) and open(target + ".stamp").read().strip() == want
    except OSError:
        return False


def build(force=False, verbose=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()
    hdr_digest = _digest(hdrs, " ".join(FLAGS))
    todo = []
    objs = []
    stamps = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        stamp = _digest([src], hdr_digest)
        objs.append(obj)
        stamps.append(stamp)
        if force or not _stamp_ok(obj, stamp):
            todo.append((src, obj, stamp))

    def cc(job):
        src, obj, stamp = job
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        with open(obj + ".stamp", "w") as f:
            f.write(stamp)
        if verbose:
            print("compiled", os.path.basename(src), file=sys.stderr)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, len(todo))) as ex:
            list(ex.map(cc, todo))
    lib_stamp = hashlib.sha256("".join(stamps).encode()).hexdigest()
    if todo or force or not _stamp_ok(LIB, lib_stamp):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        with open(LIB + ".stamp", "w") as f:
            f.write(lib_stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
