"""Parameter initialisation and flat storage for the X3D variables.

Initial values are the Keras defaults the reference relies on (it passes no initialisers):
Glorot-uniform kernels with fans computed from the TF kernel shape [k..., Cin/groups, Cout]
(fan_in = rf*Cin/g, fan_out = rf*Cout), zero biases, BN gamma=1 / beta=0 / moving_mean=0 /
moving_variance=1 (reference model.py:80-108,178-199,246-303,360-371) [TF-3p defaults].
"""
import math
from typing import Dict

import torch

from .arch import Arch, param_specs


def init_params(arch: Arch, seed: int = 0, in_channels: int = 3, dtype=torch.float32,
                device="cpu") -> Dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out = {}
    for s in param_specs(arch, in_channels):
        if s.kind in ("pw", "dw", "stem_s", "stem_t", "dense"):
            limit = math.sqrt(6.0 / (s.fan_in + s.fan_out))
            t = (torch.rand(s.shape, generator=g, dtype=torch.float64) * 2 - 1) * limit
        elif s.kind in ("gamma", "var"):
            t = torch.ones(s.shape, dtype=torch.float64)
        else:
            t = torch.zeros(s.shape, dtype=torch.float64)
        out[s.name] = t.to(dtype).to(device)
    return out


def randomize_bn_(params: Dict[str, torch.Tensor], seed: int = 1, scale: float = 0.2):
    """Perturb BN affine parameters / moving stats and biases away from their defaults.  Parity
    tests use this so that gamma/beta/bias paths are exercised (defaults of 1/0 hide sign and
    indexing errors)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    for k, v in params.items():
        r = torch.randn(v.shape, generator=g, dtype=torch.float32)
        if k.endswith("/gamma"):
            v.copy_((1.0 + scale * r).to(v.dtype))
        elif k.endswith("/beta") or k.endswith("/bias") or k.endswith("/moving_mean"):
            v.copy_((scale * r).to(v.dtype))
        elif k.endswith("/moving_variance"):
            v.copy_((1.0 + scale * r).abs().add(0.1).to(v.dtype))
    return params
