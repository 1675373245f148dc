"""MI355X-native X3D forward/backward path behind the reference's ``X3D(cfg)`` surface.

Host side mirrors reference model.py / utils.py / configs; device side is hand-written HIP for
gfx950 behind the C ABI declared in include/x3d_hip.h (libx3d_hip.so).
"""
from .config import CfgNode, get_default_config, get_config, config_path  # noqa: F401
from .arch import round_width, round_repeats, build_arch  # noqa: F401

__all__ = ["CfgNode", "get_default_config", "get_config", "config_path", "round_width",
           "round_repeats", "build_arch", "X3D"]


def __getattr__(name):
    # torch + the HIP library are only pulled in when the model is asked for
    if name == "X3D":
        from .model import X3D
        return X3D
    raise AttributeError(name)
