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
