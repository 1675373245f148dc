"""Which kernel instantiation runs behind every conv launch of a configuration -- without a GPU.

A DRY model (``X3D(cfg, device="dry")``) records the same launch list as a real one (same code: model.py
``_make_plan`` / ``_record_backward``) over address-only buffers; the library's own dispatch, in dry-run mode
(``x3d_pw_kernel_name`` / ``x3d_dw3d_kernel_name``), then names the instantiation of each launch.  Used by
tests/test_dispatch_coverage.py (every instantiation the BASELINE configurations launch at full size has an
oracle-parity case under ``-m gpu``), bench.py (roofline rows name the kernels rocprofv3 lists) and the profiling tools.
"""
import ctypes as C
from typing import Dict, List, Tuple

import torch

from . import hip
from .config import get_config
from .model import X3D

CONV_ENTRIES = ("x3d_pw_fwd", "x3d_pw_dgrad", "x3d_pw_wgrad", "x3d_pw_bwd", "x3d_dw3d_fwd", "x3d_dw3d_bwd")

# the five BASELINE.json configurations at full size: (variant, batch per GPU, T, S, dtype, training, cfg overrides)
BASELINE_CONFIGS = {
    1: ("XS", 10, 4, 160, torch.float32, False, {}),                       # one video = 10 views
    2: ("S", 32, 13, 160, torch.float32, True, {}),
    3: ("M", 64, 16, 224, torch.bfloat16, True, {}),
    4: ("L", 16, 16, 312, torch.bfloat16, True, {}),                        # yaml global batch 16 (X3D_L.yaml:24)
    5: ("XL", 30, 16, 312, torch.float16, False, {"TEST.NUM_TEMPORAL_VIEWS": 10, "TEST.NUM_SPATIAL_CROPS": 3}),
}


def describe_struct(st) -> str:
    """Shape summary of an argument struct (for messages)."""
    if isinstance(st, (hip.Dw3dFwdArgs, hip.Dw3dBwdArgs)):
        return f"N{st.N} C{st.C} T{st.T} {st.H}x{st.W} s{st.stride}"
    s = f"N{st.N} {st.Cin}->{st.Cout} T{st.T} {st.H}x{st.W}"
    if hasattr(st, "stride"):
        s += f" s{st.stride}"
    if hasattr(st, "epi"):
        s += f" epi{st.epi}"
    return s


def plan_kernels(pl) -> List[Tuple[str, str, str]]:
    """(entry point, kernel instantiation, shape summary) of every conv launch of a recorded plan, in launch order."""
    out = []
    for lst in (pl.fwd, pl.bwd):
        for i, item in enumerate(lst):
            if item is None or item[0] not in CONV_ENTRIES:
                continue
            st = pl.structs[(id(lst), i)]
            out.append((item[0], hip.kernel_name(st), describe_struct(st)))
    return out


def config_kernels(variant, batch, t, s, dtype, training, overrides=None) -> List[Tuple[str, str, str]]:
    """plan_kernels of a dry model of `variant` for `batch` clips of t x s x s."""
    flat = []
    for k, v in (overrides or {}).items():
        flat += [k, v]
    cfg = get_config(variant, flat or None)
    m = X3D(cfg, dtype=dtype, device="dry")
    pl = m._plan(batch, t, s, s, training)
    out = plan_kernels(pl)
    m.release_plans()
    return out


def baseline_kernels(index: int) -> List[Tuple[str, str, str]]:
    return config_kernels(*BASELINE_CONFIGS[index][:6], BASELINE_CONFIGS[index][6])


def kernel_set(rows) -> Dict[str, str]:
    """{kernel instantiation: first shape that dispatches it}"""
    d: Dict[str, str] = {}
    for entry, kern, shape in rows:
        d.setdefault(kern, f"{entry} {shape}")
    return d


def _struct_pointers(st, out):
    for fname, ftype in st._fields_:
        v = getattr(st, fname)
        if ftype is C.c_void_p:
            if isinstance(v, int) and v:
                out.append(v)
                if fname == "coef_fold" and v in hip.FOLDS:      # (a host struct referred to by address)
                    _struct_pointers(hip.FOLDS[v], out)
        elif isinstance(v, C.Structure):
            _struct_pointers(v, out)
        elif isinstance(v, C.Array):
            for item in v:
                if isinstance(item, C.Structure):
                    _struct_pointers(item, out)


def _pointer_values(args, st):
    """Every integer a recorded launch hands to the library that could be an address: its positional arguments and the
    pointer fields of its argument struct (nested structs and arrays of structs included: the reduce jobs of x3d_se_bnb_bwd)."""
    vals = [a for a in args if isinstance(a, int) and not isinstance(a, bool) and a]
    for a in args:
        if isinstance(a, C.Array):
            for item in a:
                if isinstance(item, C.Structure):
                    _struct_pointers(item, vals)
    if st is not None:
        _struct_pointers(st, vals)
    return vals


def gradient_writes(model, pl) -> List[Tuple[int, str, str]]:
    """(index in pl.bwd, entry point, parameter name) for every backward launch that is handed the address of a parameter's
    gradient (a view of model.flat_grads).  A launch that FINISHES a gradient later than its conv launch (the pending dW of
    a recomputed-output layer, x3d_bn_bwd_finalize_rc) shows up with its own index, which is what the data-parallel bucket
    hooks must wait for (model.forward_backward `on_stage_done`)."""
    lo = model.flat_grads.data_ptr()
    hi = lo + model.flat_grads.numel() * 4
    starts = sorted((model.grads[k].data_ptr(), k) for k in model.grads)
    out = []
    for i, item in enumerate(pl.bwd):
        if item is None or not item[2]:
            continue
        st = pl.structs.get((id(pl.bwd), i))
        for v in _pointer_values(item[2], st):
            if lo <= v < hi:
                name = [k for p, k in starts if p <= v][-1]
                out.append((i, item[0], name))
    return out


def stage_of(model, name: str) -> int:
    """Bucket of a parameter (model.grad_bucket's numbering: len(stages) = head, 0.. = residual stages, -1 = stem)."""
    if name.startswith("conv1/"):
        return -1
    if name.startswith("stages/"):
        return int(name.split("/")[1])
    return len(model.arch.stages)


# ---- read / write order of the plan-owned scratch of a backward list ---------------------------------------------------------
# Which argument of which launch READS or WRITES a scratch buffer the plan shares between launches "by stream order": the
# weight-gradient slabs, the recomputed-output operands (panel, c0, moment sums), the BatchNorm-backward coefficient tables and
# sums, the per-(n, c) coefficient table of the SE / BN_b backward.  By struct field and by argument position.
_STRUCT_ACCESS = {
    "PwBwdArgs": {"coef": ("coef", "R"), "rc_panel": ("rc_panel", "R"), "rc_c0": ("rc_c0", "R"), "rc_sums": ("rc_sums", "W"),
                  "dw_slab": ("slab", "W"), "tail_sums_c": ("bsums", "W"), "tail_sums_r": ("bsums", "W")},
    "PwWgradArgs": {"coef": ("coef", "R"), "dw_slab": ("slab", "W")},
    "PwDgradArgs": {"coef": ("coef", "R")},
    "Dw3dBwdArgs": {"coef_nc": ("coef_nc", "R"), "a_sums": ("bsums", "W")},
    "SeBnbBwdArgs": {"coef_nc": ("coef_nc", "W")},
}
_ARG_ACCESS = {
    "x3d_bn_bwd_finalize": {0: ("bsums", "R"), 4: ("coef", "W")},
    "x3d_bn_bwd_finalize_rc": {0: ("bsums", "R"), 4: ("coef", "W"), 9: ("rc_panel", "W"), 10: ("rc_c0", "W"), 12: ("rc_sums", "R"),
                               14: ("coef", "R")},
    "x3d_pw_bwd_rc_prepare": {1: ("coef", "R"), 2: ("rc_panel", "W"), 3: ("rc_c0", "W")},
    "x3d_pw_bwd_rc_finish": {0: ("rc_sums", "R"), 2: ("coef", "R")},
    "x3d_dwt_bwd": {3: ("coef", "R")},
    "x3d_stem_bwd": {3: ("coef", "R")},
    "x3d_tail_bwd": {4: ("bsums", "W"), 5: ("bsums", "W")},
    "x3d_relu_bn_bwd_reduce": {5: ("bsums", "W")},
}


def scratch_accesses(pl) -> List[Tuple[int, str, str, int, str]]:
    """(index in pl.bwd, entry point, buffer kind, address, "R" | "W") for every access of a backward launch to plan-owned scratch,
    in list order; a launch's reads come before its writes (an accumulating argument counts as a write)."""
    out = []
    for i, item in enumerate(pl.bwd):
        if item is None:
            continue
        name, _, args = item
        acc = []
        for pos, (kind, rw) in _ARG_ACCESS.get(name, {}).items():
            if pos < len(args) and isinstance(args[pos], int) and args[pos]:
                acc.append((kind, args[pos], rw))
        st = pl.structs.get((id(pl.bwd), i))
        if st is not None:
            fold = getattr(st, "coef_fold", None)
            folded = isinstance(fold, int) and bool(fold)
            for fname, (kind, rw) in _STRUCT_ACCESS.get(type(st).__name__, {}).items():
                v = getattr(st, fname)
                if isinstance(v, int) and v and not (folded and fname == "coef"):      # (with coef_fold the table is not read)
                    acc.append((kind, v, rw))
            if folded and fold in hip.FOLDS:      # the consumer derives the coefficients from the sums itself (and may publish the table)
                f = hip.FOLDS[fold]
                if f.sums:
                    acc.append(("bsums", f.sums, "R"))
                if f.coef_out:
                    acc.append(("coef", f.coef_out, "W"))
            if type(st).__name__ == "SeBnbBwdArgs":
                for job in st.reduce:
                    if job.slab:
                        acc.append(("slab", job.slab, "R"))
        for a in args:
            if isinstance(a, C.Array):                                     # x3d_dw_slab_reduce: an array of reduce jobs
                for job in a:
                    if isinstance(job, hip.DwReduceJob) and job.slab:
                        acc.append(("slab", job.slab, "R"))
        for kind, addr, rw in sorted(acc, key=lambda t: t[2]):             # "R" before "W"
            out.append((i, name, kind, addr, rw))
    return out


def scratch_hazards(pl) -> List[str]:
    """Order violations of the scratch accesses of pl.bwd: a buffer read before anything wrote it, a written value overwritten
    before a launch read it (slabs, recomputed-output operands, coefficient tables: every write must be consumed), sums written
    again after a finalize read them, a value left unread at the end of the list."""
    seq = {}
    for i, name, kind, addr, rw in scratch_accesses(pl):
        seq.setdefault((kind, addr), []).append((i, name, rw))
    bad = []
    for (kind, addr), acc in seq.items():
        pattern = "".join(rw for _, _, rw in acc)
        where = f"{kind} buffer {addr:#x}: " + " ".join(f"{rw}@{i}:{name[4:]}" for i, name, rw in acc[:8])
        if pattern[0] != "W":
            bad.append("read before any write -- " + where)
        if pattern[-1] != "R" and kind != "coef":      # (a published coefficient table may have no reader left: every consumer folds)
            bad.append("last write never read -- " + where)
        if kind == "bsums":
            if "RW" in pattern:
                bad.append("sums written after a finalize read them -- " + where)
        elif "WW" in pattern:
            bad.append("overwritten before it was read -- " + where)
    return bad
