"""Configuration tree for ``X3D(cfg)``.

The reference drives the model with a yacs ``CfgNode`` (reference configs/default.py:1-141,
merged with configs/kinetics/X3D_*.yaml at train.py:39-41).  yacs is not available here, so this
is a small attribute-access node with the same surface the hot path uses:
``get_default_config()``, ``.merge_from_file(path)``, ``.freeze()``, ``.clone()``, attribute and
item access, ``dict(cfg)``.
"""
import ast
import copy
import os

import yaml

_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


class CfgNode(dict):
    """Attribute-access dict with freeze, in the shape of yacs.config.CfgNode."""

    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f"attempted to set {name} on a frozen CfgNode")
        self[name] = value

    def __setitem__(self, key, value):
        if getattr(self, CfgNode._FROZEN, False):
            raise AttributeError(f"attempted to set {key} on a frozen CfgNode")
        super().__setitem__(key, value)

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    @staticmethod
    def _decode(value):
        # yacs decodes string leaves with literal_eval, which is how "5e-5" / "1e-5"
        # (strings to a YAML-1.1 parser) become floats.
        if isinstance(value, str):
            try:
                return ast.literal_eval(value)
            except (ValueError, SyntaxError):
                return value
        return value

    def merge_from_dict(self, other, _path=""):
        for k, v in other.items():
            full = f"{_path}.{k}" if _path else k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise ValueError(f"{full} is a leaf in the defaults but a section in the override")
                self[k].merge_from_dict(v, full)
            else:
                v = CfgNode._decode(v)
                old = self[k]
                if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
                    v = float(v)
                if isinstance(old, tuple) and isinstance(v, list):
                    v = tuple(v)
                if isinstance(old, list) and isinstance(v, tuple):
                    v = list(v)
                if old is not None and type(old) is not type(v):
                    raise ValueError(
                        f"Type mismatch ({type(old)} vs. {type(v)}) for config key: {full}")
                self[k] = v

    def merge_from_file(self, path):
        with open(path, "r") as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, kv):
        assert len(kv) % 2 == 0
        for key, val in zip(kv[0::2], kv[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node.merge_from_dict({parts[-1]: val})


def get_default_config():
    """Defaults with the keys and values of reference configs/default.py:3-140."""
    c = CfgNode()
    c.NETWORK = CfgNode(dict(
        C1_TEMP_FILTER=5, C1_CHANNELS=12, SCALE_RES2=False, WIDTH_FACTOR=1.0, DEPTH_FACTOR=1.0,
        BOTTLENECK_WIDTH_FACTOR=1.0, NUM_CLASSES=400, DROPOUT_RATE=0.0, WEIGHT_DECAY=0.00005,
        BN=dict(MOMENTUM=0.9, EPS=1e-5)))
    c.DATA = CfgNode(dict(
        FRAME_RATE=1, TEMP_DURATION=1, NUM_INPUT_CHANNELS=3, TRAIN_JITTER_SCALES=[182, 228],
        TRAIN_CROP_SIZE=112, TEST_CROP_SIZE=160, MEAN=[0.45, 0.45, 0.45],
        STD=[0.225, 0.225, 0.225]))
    c.TRAIN = CfgNode(dict(
        DATASET_SIZE=0, BATCH_SIZE=1, EPOCHS=1, OPTIMIZER="SGD", MOMENTUM=0.9, BASE_LR=0.1,
        WARMUP_EPOCHS=1, WARMUP_LR=0.01))
    c.TEST = CfgNode(dict(NUM_SPATIAL_CROPS=3, NUM_TEMPORAL_VIEWS=1, BATCH_SIZE=1))
    c.WANDB = CfgNode(dict(
        ENABLE=False, PROJECT_NAME="X3D-tf", GROUP_NAME=" ", MODE="online", TENSORBOARD=True))
    return c


def config_path(name):
    """Path of a shipped model config: name in {XS, S, M, L, XL} or 'X3D_M'."""
    name = name.upper().replace("X3D-", "").replace("X3D_", "")
    return os.path.join(_CONFIG_DIR, "kinetics", f"X3D_{name}.yaml")


def get_config(name, overrides=None, freeze=True):
    """Defaults + configs/kinetics/X3D_<name>.yaml (+ optional [key, value, ...] overrides)."""
    cfg = get_default_config()
    cfg.merge_from_file(config_path(name))
    if overrides:
        cfg.merge_from_list(list(overrides))
    if freeze:
        cfg.freeze()
    return cfg
