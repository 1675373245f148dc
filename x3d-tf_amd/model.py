"""``X3D(cfg)``: the reference's model surface (reference model.py:8-132) over hand-written HIP kernels.

Same constructor argument (the config tree), same call signature (``model(input, training=False)``
with a channels-last ``[N, T, H, W, 3]`` clip batch, fp32 probabilities out, views averaged at
inference -- model.py:113-127), same attribute tree (``conv1.conv_s``, ``stages[i].stage[j].bottleneck.a``
... -- the names the released checkpoints are keyed by), ``summary(input_shape)`` and
``load_weights(path)``.  What is different is everything underneath: activations live in NCTHW, every
op is a kernel from libx3d_hip.so, BatchNorm / ReLU / swish / SE scale never touch HBM as separate
passes (they are folded into the prologue of the consuming conv and the epilogue of the producing
one), and the backward pass is written out explicitly instead of taped.

Execution model: for one (batch, clip shape, mode) a ``_Plan`` allocates every buffer once and records
the kernel launches as pre-bound C calls; a step replays the list on the current HIP stream.  Nothing in a
replay allocates or synchronises, but a replay is NOT capture-safe as it stands: with 16-bit storage the stem
launches are re-bound to the CALLER's batch on every call (`_bind_input`: the clips are read in place, forward
AND backward -- `stem_s_wgrad` reads them again, so the caller must not overwrite the batch between the two), a
captured graph would bake that pointer in.  `options={"stem_nthwc": False}` restores the static planar input
buffer (one conversion pass per step) for graph capture; replaying as a hipGraph measured no gain.
"""
import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch

from . import hip
from .arch import Arch, BlockSpec, ParamSpec, block_prefix, build_arch, param_specs, same_pad, summary_rows
from .hip import (ACT_NONE, ACT_RELU, ACT_SWISH, EPI_ADD, EPI_ADD_STRIDED, EPI_STORE, EPI_SWISH_BWD)
from .params import init_params


# ------------------------------------------------------------------------------------------------
# thin layer objects: they own views of the flat parameter buffers and give the reference's
# attribute paths something to resolve to.  No compute happens here.
# ------------------------------------------------------------------------------------------------
class _Layer:
    def __init__(self, name):
        self.name = name

    def variables(self):
        return {k: v for k, v in vars(self).items() if isinstance(v, torch.Tensor)}


class Conv3D(_Layer):
    def __init__(self, name, kernel, bias=None, **attrs):
        super().__init__(name)
        self.kernel = kernel
        if bias is not None:
            self.bias = bias
        self.__dict__.update(attrs)


class Dense(Conv3D):
    pass


class BatchNormalization(_Layer):
    def __init__(self, name, gamma, beta, moving_mean, moving_variance, epsilon, momentum):
        super().__init__(name)
        self.gamma, self.beta = gamma, beta
        self.moving_mean, self.moving_variance = moving_mean, moving_variance
        self.epsilon, self.momentum = epsilon, momentum
        self.axis = -1


class Activation(_Layer):
    def __init__(self, kind):
        super().__init__(kind)
        self.activation = kind


class AdaptiveAvgPool3D(_Layer):
    def __init__(self, name="pool"):
        super().__init__(name)
        self.out_shape = (1, 1, 1)


class Dropout(_Layer):
    def __init__(self, rate):
        super().__init__("dropout")
        self.rate = rate


class X3D_Stem(_Layer):
    pass


class Bottleneck(_Layer):
    pass


class ResBlock(_Layer):
    pass


class ResStage(_Layer):
    pass


class _Sequential(list):
    """K.Sequential stand-in: an indexable list of layers (``layer_with_weights-i`` order)."""
    pass


# ------------------------------------------------------------------------------------------------
# Plan options: which of several launch lists -- each covered by the GPU tests -- a plan records.  The defaults are the product;
# the others document what was measured against them and serve the differential tests (X3D(cfg, options=...)).
# They are constructor arguments, not environment switches: the product path reads no X3D_* variable.  Only with
# X3D_EXPERIMENTS=1 (the A/B tools under tools/) are the historical variable names mapped onto them.
PLAN_DEFAULTS = {
    "fused_pw_bwd": True,      # x3d_pw_bwd (data + weight gradient in one launch) where it applies; False: x3d_pw_dgrad + x3d_pw_wgrad
    "pw_bwd_rc": True,         # ... in the form that recomputes the conv output algebraically instead of reading a_raw / r_raw
    "pw_bwd_rc_merge": True,   # its prepare / finish jobs ride on the BatchNorm-backward finalize launches
    "pw_bwd_rc_wide": True,    # ... also for the 48 -> 216 layer
    "stem_nthwc": True,        # the stem reads the caller's channels-last batch in place (16-bit storage)
    "stem_fused": True,        # ... and runs conv_s -> conv_t as ONE launch each way (x3d_stem_fwd / x3d_stem_bwd): no s_raw, no ds
    "shortcut_compact": True,  # strided shortcut convs whose output rows are odd (7, 39, 5 wide: one output per 4-byte load in the
                               #   gather) read an even-pixel copy of the block input instead (x3d_subsample2): dense launches
    "infer_train_plan": False,  # training-shaped launch list at inference
    "bn_fold": False,          # BatchNorm finalize inside the depthwise / tail consumer (x3d_bn_fold)
    "tail_fwd_fold": True,     # residual tail built on load by the next block's `a` conv
    "tail_fold_wst": True,     # ... also where that conv runs the weights-stationary kernel (stages 4 / 5)
    "tail_bwd_fold": True,     # Add + ReLU backward in the epilogue of the kernel that produces dy
    "stem_bwd_fold": True,     # the stem BatchNorm's backward sums in the first block's `a` backward
    "side_wgrad": False,       # unfused weight-gradient GEMMs on a side stream
    "coef_fold": True,         # the BatchNorm-backward finalize derived by its consumers where their kernels take it (no launch)
    "dw_slab": True,           # persistent fused backward kernels store per-workgroup partial weight gradients (plain stores) that
                               #   the next x3d_se_bnb_bwd launch adds up, instead of flushing them with fp32 atomics
}
_ENV_OPTIONS = {   # historical switch -> (option, value the variable's non-default setting selects)
    "X3D_NO_FUSED_PW_BWD": ("fused_pw_bwd", "1", False), "X3D_PW_BWD_RC": ("pw_bwd_rc", "0", False),
    "X3D_PW_BWD_RC_MERGE": ("pw_bwd_rc_merge", "0", False), "X3D_PW_BWD_RC_WIDE": ("pw_bwd_rc_wide", "0", False),
    "X3D_NO_STEM_NTHWC": ("stem_nthwc", "1", False), "X3D_NO_STEM_FUSED": ("stem_fused", "1", False), "X3D_NO_SHORTCUT_COMPACT": ("shortcut_compact", "1", False), "X3D_INFER_TRAIN_PLAN": ("infer_train_plan", "1", True),
    "X3D_BN_FOLD": ("bn_fold", "1", True), "X3D_NO_TAIL_FWD_FOLD": ("tail_fwd_fold", "1", False),
    "X3D_NO_TAIL_FOLD_WST": ("tail_fold_wst", "1", False), "X3D_NO_TAIL_FOLD": ("tail_bwd_fold", "1", False),
    "X3D_NO_STEM_BWD_FOLD": ("stem_bwd_fold", "1", False), "X3D_SIDE_WGRAD": ("side_wgrad", "1", True),
    "X3D_NO_DW_SLAB": ("dw_slab", "1", False), "X3D_NO_COEF_FOLD": ("coef_fold", "1", False),
}


def _experiment_options():
    if os.environ.get("X3D_EXPERIMENTS") != "1":
        return {}
    return {opt: val for var, (opt, trigger, val) in _ENV_OPTIONS.items() if os.environ.get(var) == trigger}


class _FakeBuf:
    """Stand-in for a device buffer in a DRY plan (X3D(..., device="dry")): an address range that is never touched.
    Dry plans exist so that the launch list of a full-size configuration -- and, through x3d_pw_kernel_name /
    x3d_dw3d_kernel_name, the kernel instantiation behind every launch -- can be enumerated without a GPU
    (x3d_tf_amd/dispatch.py, tests/test_dispatch_coverage.py).  Addresses are 4 KB aligned like real allocations."""
    _next = 0x7000_0000_0000

    def __init__(self, shape, dtype, ptr=None):
        self.shape, self.dtype = tuple(shape), dtype
        n = 1
        for d in self.shape:
            n *= d
        self._numel = n
        if ptr is None:
            ptr = _FakeBuf._next
            _FakeBuf._next += (n * torch.empty(0, dtype=dtype).element_size() + 4095) // 4096 * 4096 + 4096
        self._ptr = ptr

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return self._numel

    def view(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        return _FakeBuf(shape, self.dtype, self._ptr)

    def __getitem__(self, idx):   # only the `buf[:k]` the plan uses on flat scratch buffers
        assert isinstance(idx, slice) and idx.start is None and idx.step is None and len(self.shape) == 1
        return _FakeBuf((min(idx.stop, self._numel),), self.dtype, self._ptr)


class _Plan:
    """Buffers + recorded launches for one (N, T, H, W, training) configuration."""

    def __init__(self, model, n, t, h, w, training):
        self.model = model
        self.key = (n, t, h, w, bool(training))
        self.n, self.t, self.h, self.w, self.training = n, t, h, w, bool(training)
        self.fwd: List = []
        self.bwd: List = []
        self.keep: List = []           # ctypes structs / tensors that must outlive the recording
        self.lib = hip.load()
        self._zero_chunks: List = []   # (numel) fp64 accumulators carved from one flat buffer
        self._zero_views: List = []
        self.bwd_stage_marks: Dict[int, int] = {}
        self.structs: Dict = {}
        # option side_wgrad: the weight-gradient GEMMs that have no consumer before the optimizer run on
        # a SIDE stream, concurrently with the data-gradient chain (see X3D._record_backward).  Measured on X3D-M B=64
        # (r01i): 27.03 ms/step on one stream, 27.4 ms with the side stream -- the kernels already compete for the same
        # CUs and HBM, so the default is one stream.
        self.side_on = training and model.opt["side_wgrad"]
        self.side_entries = set()      # (id(list), index) of launches that go to the side stream
        self.input_slots = []          # (list, index[, argument position = 0]) of the launches that read the input batch
        self.x_cl = False              # those launches read the caller's channels-last batch in place (no planar copy)
        self.side = None               # torch.cuda.Stream, created with the first forked launch
        self._side_pending = False

    # -- allocation ------------------------------------------------------------------------------
    def act(self, *shape):
        if self.model.dry:
            return _FakeBuf(shape, self.model.dtype)
        return torch.empty(shape, dtype=self.model.dtype, device=self.model.device)

    def f32(self, *shape):
        if self.model.dry:
            return _FakeBuf(shape, torch.float32)
        return torch.empty(shape, dtype=torch.float32, device=self.model.device)

    def acc64(self, *shape):
        """fp64 accumulator zeroed at the start of every step (carved later from one flat buffer)."""
        numel = 1
        for s in shape:
            numel *= s
        self._zero_chunks.append((numel, shape))
        return len(self._zero_chunks) - 1

    def finalize_acc(self):
        total = sum(c[0] for c in self._zero_chunks)
        self.zero_buf = torch.zeros(max(total, 1), dtype=torch.float64, device=self.model.device)
        off = 0
        for numel, shape in self._zero_chunks:
            self._zero_views.append(self.zero_buf[off:off + numel].view(shape))
            off += numel

    # -- recording -------------------------------------------------------------------------------
    def rec(self, lst, name, *args):
        fn = getattr(self.lib, name)
        if fn.argtypes is not None and len(args) + 1 != len(fn.argtypes):   # (+ the stream): caught when recording, dry plans too
            raise hip.X3DHipError(f"{name}: recorded with {len(args)} arguments, the C ABI takes {len(fn.argtypes) - 1} + stream")
        conv = []
        for a in args:   # remember the argument struct of this launch (profiling tools read shapes from it)
            st = a if isinstance(a, C.Structure) else (a[1] if isinstance(a, tuple) and len(a) > 1 and isinstance(a[1], C.Structure) else None)
            if st is not None:
                self.structs[(id(lst), len(lst))] = st
        for a in args:
            if isinstance(a, (torch.Tensor, _FakeBuf)):
                self.keep.append(a)
                conv.append(a.data_ptr())
            elif isinstance(a, C.Structure):
                self.keep.append(a)
                conv.append(C.byref(a))
            else:
                conv.append(a)
        lst.append((name, fn, tuple(conv)))

    # -- side stream ------------------------------------------------------------------------------
    def rec_side(self, lst, name, *args):
        """Record a launch for the side stream: it starts after everything recorded before it and is waited for by
        the next rec_join().  Only for launches whose outputs nothing reads before that join."""
        self.rec(lst, name, *args)
        if self.side_on:
            self.side_entries.add((id(lst), len(lst) - 1))

    def rec_join(self, lst):
        """The main stream waits here for every launch forked so far (no-op when none is pending)."""
        if self.side_on:
            lst.append(("side_join", self._join, ()))

    def _join(self, stream):
        if self._side_pending:
            torch.cuda.current_stream().wait_event(self._ev_join)
            self._side_pending = False
        return 0

    def wrap_side(self, lst):
        """After the placeholders are resolved: replace the marked launches by fork wrappers."""
        if not self.side_on:
            return
        for i, (name, fn, args) in enumerate(lst):
            if (id(lst), i) in self.side_entries:
                lst[i] = (name, self._forked(fn), args)

    def _forked(self, fn):
        def launch(*a):
            if self.side is None:
                self.side = torch.cuda.Stream(device=self.model.device)
                self._ev_fork = torch.cuda.Event()
                self._ev_join = torch.cuda.Event()
            main = torch.cuda.current_stream()
            self._ev_fork.record(main)
            self.side.wait_event(self._ev_fork)
            rc = fn(*a[:-1], self.side.cuda_stream)
            self._ev_join.record(self.side)
            self._side_pending = True
            return rc
        return launch

    def run(self, lst, start=0, stop=None):
        if self.model.dry:
            raise hip.X3DHipError("a dry plan records launches; it cannot run (no CPU fallback for the hot path)")
        s = torch.cuda.current_stream().cuda_stream
        for name, fn, args in lst[start:stop]:
            st = fn(*args, s)
            if st != 0:
                hip.check(st, name)


def _p(t):
    return None if t is None else t.data_ptr()


class X3D:
    """Constructs the X3D model from the model configurations (reference model.py:8-111).

    Args:
        cfg: config tree (x3d_tf_amd.config.CfgNode or anything exposing the same attributes).
        dtype: activation storage type on the GPU: torch.float32, torch.bfloat16 or torch.float16 (weights,
            statistics and accumulation stay fp32).  float16 is the reference's reduced-precision mode (Keras
            mixed_float16, utils.py:176-192: fp16 compute, fp32 variables, loss scaling in training --
            train.Trainer's `loss_scale`); bfloat16 needs no loss scaling and is the benchmark's default.
        device: a CUDA/HIP device.  There is no CPU path.  The one exception is device="dry": the model is built on
            the host with address-only stand-ins for the activation buffers, plans can be RECORDED (to enumerate
            launches and their kernel instantiations, x3d_tf_amd/dispatch.py) and every attempt to run one raises.
        seed: seed of the Glorot-uniform initialisation.
        options: overrides of PLAN_DEFAULTS (which launch list a plan records; differential tests and A/B tools).
    """

    def __init__(self, cfg, dtype=torch.float32, device="cuda", seed: int = 0, in_channels: int = 3, options: Optional[dict] = None):
        self.cfg = cfg
        self.opt = dict(PLAN_DEFAULTS, **_experiment_options())
        for k, v in (options or {}).items():
            if k not in PLAN_DEFAULTS:
                raise ValueError(f"unknown plan option {k!r} (known: {sorted(PLAN_DEFAULTS)})")
            self.opt[k] = bool(v)
        self.arch: Arch = build_arch(cfg)
        self.num_classes = self.arch.num_classes
        self._num_preds = self.arch.num_preds
        self._bn_cfg = cfg.NETWORK.BN
        self.dtype = dtype
        self.in_channels = in_channels
        hip.dtype_code(dtype)
        self.dry = (device == "dry")
        if self.dry:
            self.device = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                raise hip.X3DHipError("X3D needs an MI355X: the HIP path has no CPU fallback")
            self.device = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        hip.load()
        self._plans: Dict = {}
        self._build_params(seed)
        self._build_panels()
        self._build_layers()
        self._dropout_mask_override = None
        self.last_loss = None
        self._fuse_pw_bwd = self.opt["fused_pw_bwd"]
        self._rc_pw_bwd = self.opt["pw_bwd_rc"]
        self._stats_r = int(hip.load().x3d_stats_replicas())

    # ---------------------------------------------------------------------------------------------
    # parameters: one flat fp32 buffer (trainable first, then BN moving statistics), one flat
    # gradient buffer and one flat momentum buffer -- a single optimizer launch and contiguous
    # all-reduce buckets.
    # ---------------------------------------------------------------------------------------------
    def _build_params(self, seed):
        specs = param_specs(self.arch, self.in_channels)
        self.specs: Dict[str, ParamSpec] = {s.name: s for s in specs}
        order = [s for s in specs if s.trainable] + [s for s in specs if not s.trainable]
        init = init_params(self.arch, seed, self.in_channels)

        def numel(s):
            n = 1
            for d in s.shape:
                n *= d
            return n

        # pad every tensor to a multiple of 4 floats so views stay 16-byte aligned
        self._offsets = {}
        off = 0
        for s in order:
            self._offsets[s.name] = off
            off += (numel(s) + 3) // 4 * 4
            if s.trainable:
                self.n_trainable_flat = off
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(self.n_trainable_flat, dtype=torch.float32, device=self.device)
        self.flat_velocity = torch.zeros(self.n_trainable_flat, dtype=torch.float32, device=self.device)
        l2 = torch.zeros(self.n_trainable_flat, dtype=torch.uint8)
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        for s in order:
            o, n = self._offsets[s.name], numel(s)
            self.params[s.name] = self.flat_params[o:o + n].view(s.shape)
            self.params[s.name].copy_(init[s.name])
            if s.trainable:
                self.grads[s.name] = self.flat_grads[o:o + n].view(s.shape)
                if s.l2:
                    l2[o:o + n] = 1
        self.l2_mask = l2.to(self.device)
        self.param_order = [s.name for s in order]
        self.n_params = sum(numel(s) for s in order)
        self.n_trainable = sum(numel(s) for s in order if s.trainable)

    def _build_panels(self):
        """bf16 LDS-image panels of every pointwise-conv weight (x3d_pw_pack_weights): refreshed by one launch
        at the start of each forward so the GEMM workgroups copy their weight rows instead of converting them."""
        self._panels: Dict[str, tuple] = {}
        self._panel_table = None
        if self.dtype not in (torch.bfloat16, torch.float16):
            return
        lib = hip.load()
        names = [s.name for s in self.specs.values() if s.kind == "pw" and
                 (s.name.endswith(("/a/kernel", "/c/kernel", "/residual/kernel")) or s.name.startswith("conv5/"))]
        sizes = []
        for nm in names:
            cout, cin = self.specs[nm].shape
            sizes.append((lib.x3d_pw_panel_elems(cout, cin), lib.x3d_pw_panel_elems(cin, cout)))
        total = sum(a + b for a, b in sizes)
        self._panel_buf = torch.zeros(total, dtype=self.dtype, device=self.device)
        items = (hip.PwPackItem * len(names))()
        off = 0
        for i, (nm, (nf, nd)) in enumerate(zip(names, sizes)):   # panel sizes are multiples of 8 elements: 16-B aligned
            fp, dp = self._panel_buf[off:off + nf], self._panel_buf[off + nf:off + nf + nd]
            off += nf + nd
            cout, cin = self.specs[nm].shape
            items[i] = hip.PwPackItem(self.params[nm].data_ptr(), fp.data_ptr(), dp.data_ptr(), cout, cin)
            self._panels[nm] = (fp, dp)
        self._panel_table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
        self._n_panels = len(names)

    def _pack_panels(self):
        if self._panel_table is not None:
            hip.call("x3d_pw_pack_weights", self._panel_table.data_ptr(), self._n_panels, hip.dtype_code(self.dtype))

    def _wp(self, name, dgrad=False):
        pr = self._panels.get(name)
        return None if pr is None else pr[1 if dgrad else 0].data_ptr()

    def _bn(self, prefix):
        p = self.params
        return BatchNormalization(prefix, p[f"{prefix}/gamma"], p[f"{prefix}/beta"],
                                  p[f"{prefix}/moving_mean"], p[f"{prefix}/moving_variance"],
                                  self.arch.bn_eps, self.arch.bn_momentum)

    def _build_layers(self):
        p, a = self.params, self.arch
        self.conv1 = X3D_Stem("conv_1")
        self.conv1.conv_s = Conv3D("conv_s", p["conv1/conv_s/kernel"], kernel_size=(1, 3, 3), strides=(1, 2, 2))
        self.conv1.conv_t = Conv3D("conv_t", p["conv1/conv_t/kernel"], kernel_size=(a.c1_temp_filter, 1, 1),
                                   strides=(1, 1, 1), groups=a.c1)
        self.conv1.bn = self._bn("conv1/bn")
        self.conv1.relu = Activation("relu")
        self.stages = []
        for si, st in enumerate(a.stages):
            stage = ResStage(f"res_stage_{si + 2}")
            stage._inner_channels = st.inner
            stage.stage = _Sequential()
            for b in st.blocks:
                pre = block_prefix(b)
                q = f"{pre}/bottleneck"
                rb = ResBlock(f"ResBlock_{b.global_index - 1}")
                rb.in_channels, rb.inner_channels, rb.out_channels = b.cin, b.inner, b.cout
                if b.has_shortcut_conv:
                    rb.residual = Conv3D("residual", p[f"{pre}/residual/kernel"], kernel_size=(1, 1, 1),
                                         strides=(1, b.stride, b.stride))
                    rb.bn_r = self._bn(f"{pre}/bn_r")
                bt = Bottleneck("bottleneck")
                bt.block_index = b.global_index
                bt.a = Conv3D("a", p[f"{q}/a/kernel"], kernel_size=(1, 1, 1), strides=(1, 1, 1))
                bt.bn_a = self._bn(f"{q}/bn_a")
                bt.relu = Activation("relu")
                bt.b = Conv3D("b", p[f"{q}/b/kernel"], kernel_size=(3, 3, 3), strides=(1, b.stride, b.stride),
                              groups=b.inner)
                bt.bn_b = self._bn(f"{q}/bn_b")
                bt.swish = Activation("swish")
                if b.has_se:
                    bt.se_pool = AdaptiveAvgPool3D("se_pool")
                    bt.se_fc1 = Conv3D("se_fc1", p[f"{q}/se_fc1/kernel"], p[f"{q}/se_fc1/bias"])
                    bt.se_fc2 = Conv3D("se_fc2", p[f"{q}/se_fc2/kernel"], p[f"{q}/se_fc2/bias"])
                bt.c = Conv3D("c", p[f"{q}/c/kernel"], kernel_size=(1, 1, 1), strides=(1, 1, 1))
                bt.bn_c = self._bn(f"{q}/bn_c")
                rb.bottleneck = bt
                rb.add_op = Activation("add")
                rb.relu = Activation("relu")
                stage.stage.append(rb)
            self.stages.append(stage)
        self.conv5 = _Sequential([
            Conv3D("conv_5", p["conv5/layer_with_weights-0/kernel"], kernel_size=(1, 1, 1)),
            self._bn("conv5/layer_with_weights-1"), Activation("relu")])
        self.pool5 = AdaptiveAvgPool3D("pool_5")
        self.fc1 = Conv3D("fc_1", p["fc1/kernel"], kernel_size=(1, 1, 1))
        self.dropout = Dropout(a.dropout_rate)
        self.fc2 = Dense("fc_2", p["fc2/kernel"], p["fc2/bias"])
        self.softmax = Activation("softmax")

    # ---------------------------------------------------------------------------------------------
    def summary(self, input_shape, print_fn=print):
        """Same table as the reference's ``X3D.summary`` (model.py:129-132, models/X3D-*/X3D_*.txt)."""
        t, h, w, c = input_shape
        rows = summary_rows(self.arch, t, h, w, c)
        lines = ['Model: "X3D"', "_" * 65, f"{'Layer (type)':<29}{'Output Shape':<26}{'Param #':<10}", "=" * 65,
                 f"{'input_1 (InputLayer)':<29}{str([(None, t, h, w, c)]):<26}{0:<10}", "_" * 65]
        kinds = {"conv_1": "X3D_Stem", "conv_5": "Sequential", "pool_5": "AdaptiveAvgPool3D", "fc_1": "Conv3D",
                 "dropout": "Dropout", "fc_2": "Dense"}
        for name, shp, n in rows:
            kind = kinds.get(name, "ResStage")
            lines += [f"{name + ' (' + kind + ')':<29}{str((None,) + tuple(shp)):<26}{n:<10}", "_" * 65]
        lines[-1] = "=" * 65
        lines += [f"Total params: {self.n_params:,}", f"Trainable params: {self.n_trainable:,}",
                  f"Non-trainable params: {self.n_params - self.n_trainable:,}", "_" * 65]
        text = "\n".join(lines)
        if print_fn:
            print_fn(text)
        return text

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.params.items()}

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.params if k not in sd]
        if strict and missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for k, v in sd.items():
            if k in self.params:
                if tuple(v.shape) != tuple(self.params[k].shape):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} vs {tuple(self.params[k].shape)}")
                self.params[k].copy_(v)
            elif strict:
                raise KeyError(f"unexpected parameter {k}")

    def load_weights(self, path, expect_partial=True, optimizer=None):
        """Keras-style ``load_weights`` on a TF tensor-bundle checkpoint prefix or directory
        (reference train.py:131-143, eval.py:78-81).  optimizer: "sgd" | "adam" -- the branch that will use the
        optimizer slots (slots written by the other branch are left at zero); None installs what the bundle holds."""
        from .checkpoint import load_tf_checkpoint
        return load_tf_checkpoint(self, path, expect_partial=expect_partial, optimizer=optimizer)

    def _claim_slots(self, kind):
        """The slot buffers hold state of ONE optimizer branch (`slot_kind`: set by load_weights, then by the first update).
        An update of the other branch starts from zero slots, as a freshly built Keras optimizer does."""
        have = getattr(self, "slot_kind", None)
        if have is not None and have != kind:
            self.flat_velocity.zero_()
            if getattr(self, "flat_second", None) is not None:
                self.flat_second.zero_()
        self.slot_kind = kind

    def save_weights(self, prefix, optimizer_hyper=None, optimizer="sgd"):
        from .checkpoint import save_tf_checkpoint
        return save_tf_checkpoint(self, prefix, optimizer_hyper, optimizer)

    # ---------------------------------------------------------------------------------------------
    # plan construction
    # ---------------------------------------------------------------------------------------------
    MAX_PLANS = 3   # a plan owns every activation / gradient / scratch buffer of its shape (X3D-M, B = 64, bf16: ~16 GB)

    def _plan(self, n, t, h, w, training) -> _Plan:
        key = (n, t, h, w, bool(training))
        pl = self._plans.pop(key, None)
        if pl is None:
            self.release_plans(keep=self.MAX_PLANS - 1)
            pl = self._make_plan(n, t, h, w, bool(training))
        self._plans[key] = pl           # most recently used last
        return pl

    def release_plans(self, keep: int = 0):
        """Drop all but the `keep` most recently used plans (their buffers return to the allocator once no launch that
        uses them is pending: the stream is synchronised first)."""
        if len(self._plans) <= keep:
            return
        if not self.dry:
            torch.cuda.synchronize(self.device)
        for key in list(self._plans)[:len(self._plans) - keep]:
            del self._plans[key]

    def _make_infer_plan(self, n, t, h, w) -> _Plan:
        """The forward pass at training=False (reference model.py:113-127; eval.py:83-89) as its own launch list.

        With the moving statistics every BatchNorm is a per-channel affine known before the first kernel, so nothing
        waits for batch statistics and the training plan's materialised intermediates disappear:
          stem     conv_s -> conv_t with BN + ReLU in its epilogue (x3d_dwt_fwd out_scale_shift): no raw t tensor, no tail
          block    a (raw) -> b (BN_a + ReLU on load; SE squeeze in the epilogue) -> [SE MLP] -> [strided shortcut conv (raw)]
                   -> c with BN_b * gate -> swish on load and  relu(bn_c(acc) + shortcut)  in its epilogue
                   (x3d_pw_fwd out_scale_shift / out_add / out_add_scale_shift): no c_raw, no residual-tail pass
          head     conv5 (raw) -> pool of relu(bn(.)) -> fc1 -> fc2 -> softmax -> view mean
        3 launches per block (4 with SE, +1 for a stage's first block) instead of 4-6, and 2 tensor passes of Cout*P less
        per block.  The per-layer coefficients still come from ONE batched launch at the head of the list (they depend on
        the parameters only, but parameters may change between calls).  Activation buffers are shared between blocks
        (two block outputs ping-pong; one a / b / shortcut scratch each), so a 30-view X3D-XL plan holds ~3 GB, not ~30."""
        a, p = self.arch, self.params
        pl = _Plan(self, n, t, h, w, False)
        dt = hip.dtype_code(self.dtype)
        eps = a.bn_eps
        F = pl.fwd
        if n % a.num_preds:
            raise ValueError(f"inference batch {n} is not a multiple of views*crops={a.num_preds} "
                             "(reference model.py:125)")

        class BNBuf:
            pass

        pl.bn_eval_items = []

        def bn_coef(prefix, c):
            b = BNBuf()
            b.prefix, b.c = prefix, c
            b.ss, b.mi = pl.f32(c, 2), pl.f32(c, 2)
            pl.bn_eval_items.append(hip.BnEvalItem(_p(p[f"{prefix}/gamma"]), _p(p[f"{prefix}/beta"]),
                                                   _p(p[f"{prefix}/moving_mean"]), _p(p[f"{prefix}/moving_variance"]),
                                                   _p(b.ss), _p(b.mi), c))
            return b

        F.append(None)   # slot 0: x3d_bn_eval_coef_batched, filled in once every BN layer is known
        # ---- geometry first: the shared buffers are sized for their largest user ----------------------------------
        h1, w1 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        geo, hh, ww = [], h1, w1
        for b in a.blocks:
            ho, wo = same_pad(hh, 3, b.stride)[0], same_pad(ww, 3, b.stride)[0]
            geo.append((hh, ww, ho, wo))
            hh, ww = ho, wo
        numel_y = max([n * a.c1 * t * h1 * w1] + [n * b.cout * t * g[2] * g[3] for b, g in zip(a.blocks, geo)])
        numel_a = max(n * b.inner * t * g[0] * g[1] for b, g in zip(a.blocks, geo))
        numel_b = max(n * b.inner * t * g[2] * g[3] for b, g in zip(a.blocks, geo))
        numel_r = max([1] + [n * b.cout * t * g[2] * g[3] for b, g in zip(a.blocks, geo) if b.has_shortcut_conv])
        ybuf = [pl.act(numel_y), pl.act(numel_y)]
        abuf, bbuf, rbuf = pl.act(numel_a), pl.act(numel_b), pl.act(numel_r)

        def view(buf, *shape):
            numel = 1
            for d in shape:
                numel *= d
            return buf[:numel].view(*shape)

        # ---- input + stem ------------------------------------------------------------------------------------------
        pl.x_in = None
        # 16-bit storage: the stem's matrix-core kernels read the caller's channels-last batch in place (x3d_hip.h K1)
        pl.x_cl = bool(pl.lib.x3d_stem_s_nthwc_supported(self.in_channels, w, a.c1, dt)) and self.opt["stem_nthwc"]
        pl.x = None if pl.x_cl else pl.act(n, self.in_channels, t, h, w)
        pl.stem_fused = self._stem_fused(pl, n, t, h, w, dt)
        pl.y0 = view(ybuf[0], n, a.c1, t, h1, w1)
        pl.bn1 = bn_coef("conv1/bn", a.c1)
        pl.input_slots.append((F, len(F)))     # (the launch reads the caller's batch in place when pl.x_cl: _bind_input)
        if pl.stem_fused:   # conv_s -> conv_t -> BN + ReLU in one launch: the conv_s output stays on chip (reference model.py:202-208)
            pl.s_raw = None
            pl.rec(F, "x3d_stem_fwd", pl.x, p["conv1/conv_s/kernel"], p["conv1/conv_t/kernel"], pl.y0, None, pl.bn1.ss, ACT_RELU,
                   n, self.in_channels, t, h, w, a.c1, a.c1_temp_filter, dt, 1)
        else:
            pl.s_raw = view(abuf, n, a.c1, t, h1, w1)     # conv_s output: dead once conv_t has run, shares the `a` scratch
            pl.rec(F, "x3d_stem_s_fwd", pl.x, p["conv1/conv_s/kernel"], pl.s_raw, n, self.in_channels, t, h, w, a.c1, dt, int(pl.x_cl))
            pl.rec(F, "x3d_dwt_fwd", pl.s_raw, p["conv1/conv_t/kernel"], pl.y0, None, pl.bn1.ss, ACT_RELU, n, a.c1, t, h1 * w1,
                   a.c1_temp_filter, dt)
        # ---- residual stages ---------------------------------------------------------------------------------------
        x_cur, cur = pl.y0, 0
        pl.blocks = []
        for b, (hh, ww, ho, wo) in zip(a.blocks, geo):
            pre = block_prefix(b)
            q = f"{pre}/bottleneck"
            P_out = t * ho * wo

            class B:
                pass

            B.spec, B.x, B.hh, B.ww, B.ho, B.wo = b, x_cur, hh, ww, ho, wo
            B.a_raw = view(abuf, n, b.inner, t, hh, ww)
            B.b_raw = view(bbuf, n, b.inner, t, ho, wo)
            B.y = view(ybuf[1 - cur], n, b.cout, t, ho, wo)
            B.bn_a, B.bn_b, B.bn_c = bn_coef(f"{q}/bn_a", b.inner), bn_coef(f"{q}/bn_b", b.inner), bn_coef(f"{q}/bn_c", b.cout)
            B.pool = pl.acc64(n, b.inner) if b.has_se else None
            B.gate = pl.f32(n, b.inner) if b.has_se else None
            B.hidden = pl.f32(n, b.se_width) if b.has_se else None
            sa = hip.PwFwdArgs(_p(x_cur), _p(p[f"{q}/a/kernel"]), _p(B.a_raw), None, None, None, ACT_NONE, n, b.cin,
                               b.inner, t, hh, ww, 1, dt)
            sa.w_panel = self._wp(f"{q}/a/kernel")
            B.sa = sa
            pl.rec(F, "x3d_pw_fwd", sa)
            sb = hip.Dw3dFwdArgs(_p(B.a_raw), _p(p[f"{q}/b/kernel"]), _p(B.b_raw), _p(B.bn_a.ss), ACT_RELU, None, None,
                                 n, b.inner, t, hh, ww, b.stride, dt)
            B.sb = sb
            pl.rec(F, "x3d_dw3d_fwd", ("dwstats", sb, None, B.pool))
            if b.has_se:
                pl.rec(F, "x3d_se_fwd", ("acc", B.pool), float(P_out), B.bn_b.ss, p[f"{q}/se_fc1/kernel"],
                       p[f"{q}/se_fc1/bias"], p[f"{q}/se_fc2/kernel"], p[f"{q}/se_fc2/bias"], B.gate, B.hidden, n,
                       b.inner, b.se_width)
            if b.has_shortcut_conv:
                B.r_raw = view(rbuf, n, b.cout, t, ho, wo)
                B.bn_r = bn_coef(f"{pre}/bn_r", b.cout)
                if b.stride == 2 and wo % 2 == 1 and self.opt["shortcut_compact"]:
                    # the pixels the strided conv samples, copied once: the conv itself is a dense launch on them
                    B.xs = pl.act(n, b.cin, t, ho, wo)
                    pl.rec(F, "x3d_subsample2", x_cur, B.xs, n * b.cin * t, hh, ww, dt)
                    sr = hip.PwFwdArgs(_p(B.xs), _p(p[f"{pre}/residual/kernel"]), _p(B.r_raw), None, None, None, ACT_NONE,
                                       n, b.cin, b.cout, t, ho, wo, 1, dt)
                else:
                    sr = hip.PwFwdArgs(_p(x_cur), _p(p[f"{pre}/residual/kernel"]), _p(B.r_raw), None, None, None, ACT_NONE,
                                       n, b.cin, b.cout, t, hh, ww, b.stride, dt)
                sr.w_panel = self._wp(f"{pre}/residual/kernel")
                B.sr = sr
                pl.rec(F, "x3d_pw_fwd", sr)
                add, add_ss = B.r_raw, B.bn_r.ss
            else:
                B.r_raw, B.bn_r = None, None
                add, add_ss = x_cur, None
            # c: BN_b * gate -> swish on load; relu(bn_c(acc) + shortcut) on the accumulators
            sc = hip.PwFwdArgs(_p(B.b_raw), _p(p[f"{q}/c/kernel"]), _p(B.y), None, _p(B.bn_b.ss), _p(B.gate),
                               ACT_SWISH, n, b.inner, b.cout, t, ho, wo, 1, dt, self._wp(f"{q}/c/kernel"),
                               out_scale_shift=_p(B.bn_c.ss), out_add=_p(add), out_add_scale_shift=_p(add_ss), out_act=ACT_RELU)
            B.sc = sc
            pl.rec(F, "x3d_pw_fwd", sc)
            pl.blocks.append(B)
            x_cur, cur = B.y, 1 - cur
        # ---- head --------------------------------------------------------------------------------------------------
        hh, ww = geo[-1][2], geo[-1][3]
        c_last, c5 = a.stages[-1].cout, a.conv5_out
        P5 = t * hh * ww
        pl.P5, pl.h5, pl.w5 = P5, hh, ww
        pl.y_last = x_cur
        pl.c5_raw = view(abuf, n, c5, t, hh, ww) if n * c5 * P5 <= numel_a else pl.act(n, c5, t, hh, ww)
        pl.bn5 = bn_coef("conv5/layer_with_weights-1", c5)
        s5 = hip.PwFwdArgs(_p(x_cur), _p(p["conv5/layer_with_weights-0/kernel"]), _p(pl.c5_raw), None, None, None,
                           ACT_NONE, n, c_last, c5, t, hh, ww, 1, dt)
        s5.w_panel = self._wp("conv5/layer_with_weights-0/kernel")
        pl.s5 = s5
        pl.rec(F, "x3d_pw_fwd", s5)
        pl.pooled = pl.f32(n, c5)
        pl.h1 = pl.f32(n, a.fc1_out)
        pl.logits = pl.f32(n, a.num_classes)
        pl.probs = pl.f32(n, a.num_classes)
        pl.drop_mask, pl.drop_scale = None, 1.0
        pl.rec(F, "x3d_pool_fwd", pl.c5_raw, pl.bn5.ss, pl.pooled, n, c5, P5, dt)
        pl.rec(F, "x3d_dense_fwd", pl.pooled, None, 1.0, p["fc1/kernel"], None, pl.h1, ACT_RELU, n, c5, a.fc1_out)
        pl.rec(F, "x3d_dense_fwd", pl.h1, None, 1.0, p["fc2/kernel"], p["fc2/bias"], pl.logits, ACT_NONE, n, a.fc1_out,
               a.num_classes)
        pl.rec(F, "x3d_softmax_xent", pl.logits, None, pl.probs, None, None, 1.0, n, a.num_classes)
        pl.out = pl.f32(n // a.num_preds, a.num_classes)
        pl.rec(F, "x3d_view_mean", pl.probs, pl.out, n // a.num_preds, a.num_preds, a.num_classes)
        items = (hip.BnEvalItem * len(pl.bn_eval_items))(*pl.bn_eval_items)
        pl.bn_eval_table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
        F[0] = ("x3d_bn_eval_coef_batched", pl.lib.x3d_bn_eval_coef_batched,
                (pl.bn_eval_table.data_ptr(), len(pl.bn_eval_items), float(eps)))
        pl.folds = []
        pl.finalize_acc()
        self._resolve(pl, pl.fwd)
        return pl

    def _make_plan(self, n, t, h, w, training) -> _Plan:
        if not training and not self.opt["infer_train_plan"]:
            return self._make_infer_plan(n, t, h, w)
        a, p = self.arch, self.params
        pl = _Plan(self, n, t, h, w, training)
        dt = hip.dtype_code(self.dtype)
        eps, mom = a.bn_eps, a.bn_momentum
        F = pl.fwd

        class BNBuf:
            pass

        def bn_bufs(prefix, c):
            b = BNBuf()
            b.prefix, b.c = prefix, c
            b.ss = pl.f32(c, 2)
            b.mi = pl.f32(c, 2)
            # forward statistics: the library's replicated layout (x3d_stats_replicas copies, x3d_stats_stride apart)
            b.stats = pl.acc64(self._stats_r * int(pl.lib.x3d_stats_stride(c))) if training else None
            b.bsums = pl.acc64(c, 2) if training else None
            b.coef = pl.f32(c, 4) if training else None
            return b

        def bn_finish(b, count):
            """after the producer kernel: turn statistics (training) or moving stats (inference) into scale/shift"""
            g, be = p[f"{b.prefix}/gamma"], p[f"{b.prefix}/beta"]
            mm, mv = p[f"{b.prefix}/moving_mean"], p[f"{b.prefix}/moving_variance"]
            if training:
                pl.rec(F, "x3d_bn_finalize", ("acc", b.stats), float(count), g, be, mm, mv, float(eps), float(mom), 1,
                       b.ss, b.mi, b.c)
            else:   # inference: every layer's coefficients in ONE launch at the head of the forward list (below)
                pl.bn_eval_items.append(hip.BnEvalItem(_p(g), _p(be), _p(mm), _p(mv), _p(b.ss), _p(b.mi), b.c))

        # option bn_fold (training): the finalize of a BatchNorm whose consumer has one channel per
        # workgroup (depthwise conv, residual tail) runs inside that consumer (x3d_bn_fold) -- 57 launches fewer per
        # X3D-M step, worth 0.08 ms with single-copy statistics.  With the replicated accumulators every consumer
        # workgroup would have to sum 32 copies first, so the separate x3d_bn_finalize launches are the default.
        fold_on = training and self.opt["bn_fold"]
        pl.folds = []

        def bn_fold(b, count):
            f = hip.BnFold(None, float(count), _p(p[f"{b.prefix}/gamma"]), _p(p[f"{b.prefix}/beta"]),
                           _p(p[f"{b.prefix}/moving_mean"]), _p(p[f"{b.prefix}/moving_variance"]), float(eps), float(mom), 1,
                           _p(b.ss), _p(b.mi))
            pl.folds.append((f, b.stats))   # stats pointer resolved with the other fp64 accumulators
            pl.keep.append(f)
            return f

        pl.bn_eval_items = []
        if not training:
            F.append(None)   # slot 0: x3d_bn_eval_coef_batched, filled in once every BN layer is known
        # ---- input + stem --------------------------------------------------------------------
        pl.x_in = None  # bound at run time (NTHWC user tensor)
        # 16-bit storage: the stem's matrix-core kernels read the caller's channels-last batch in place (x3d_hip.h K1)
        pl.x_cl = bool(pl.lib.x3d_stem_s_nthwc_supported(self.in_channels, w, a.c1, dt)) and self.opt["stem_nthwc"]
        pl.x = None if pl.x_cl else pl.act(n, self.in_channels, t, h, w)
        h1, w1 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        # conv_s -> conv_t as one launch each way where the fused kernels take the shape: the conv_s output (616 MB at the
        # headline's size) and its gradient never exist in HBM (reference model.py:202-206: nothing between the two convs)
        pl.stem_fused = self._stem_fused(pl, n, t, h, w, dt)
        pl.s_raw = None if pl.stem_fused else pl.act(n, a.c1, t, h1, w1)
        pl.t_raw = pl.act(n, a.c1, t, h1, w1)
        pl.y0 = pl.act(n, a.c1, t, h1, w1)
        pl.bn1 = bn_bufs("conv1/bn", a.c1)
        pl.input_slots.append((F, len(F)))     # (the launch reads the caller's batch in place when pl.x_cl: _bind_input)
        if pl.stem_fused:
            pl.rec(F, "x3d_stem_fwd", pl.x, p["conv1/conv_s/kernel"], p["conv1/conv_t/kernel"], pl.t_raw,
                   ("acc", pl.bn1.stats) if training else None, None, ACT_NONE, n, self.in_channels, t, h, w, a.c1,
                   a.c1_temp_filter, dt, 1)
        else:
            pl.rec(F, "x3d_stem_s_fwd", pl.x, p["conv1/conv_s/kernel"], pl.s_raw, n, self.in_channels, t, h, w, a.c1, dt, int(pl.x_cl))
            pl.rec(F, "x3d_dwt_fwd", pl.s_raw, p["conv1/conv_t/kernel"], pl.t_raw,
                   ("acc", pl.bn1.stats) if training else None, None, ACT_NONE, n, a.c1, t, h1 * w1, a.c1_temp_filter, dt)
        if fold_on:
            pl.rec(F, "x3d_tail_fwd_bn", pl.t_raw, bn_fold(pl.bn1, n * t * h1 * w1), None, None, pl.y0, n, a.c1, t * h1 * w1, dt)
        else:
            bn_finish(pl.bn1, n * t * h1 * w1)
            stem_tail = (pl.t_raw, pl.bn1.ss, None, None, pl.y0, a.c1, t * h1 * w1)   # deferred to its first reader (below)

        # ---- residual stages -------------------------------------------------------------------
        # The residual tail of a block (y = relu(bn_c(c) + shortcut), reference model.py:381-392) is DEFERRED to the first
        # reader of y: the next block's `a` conv (or conv5) builds y on load and stores it (x3d_pw_fwd in_add / in_store) where
        # that form exists and does not cost the layer its weights-stationary kernel; otherwise x3d_tail_fwd runs first.
        pending = {"tail": None if fold_on or not training else stem_tail}
        if not training and not fold_on:
            pl.rec(F, "x3d_tail_fwd", pl.t_raw, pl.bn1.ss, None, None, pl.y0, n, a.c1, t * h1 * w1, dt)
        fold_fwd = training and not fold_on and self.opt["tail_fwd_fold"]

        def fold_pending_tail(st):
            """st: the x3d_pw_fwd arguments of the conv that reads the pending block output first."""
            tl, pending["tail"] = pending["tail"], None
            if tl is None:
                return st
            c_raw, c_ss, shortcut, r_ss, y, cout_, p_out_ = tl
            ft = hip.PwFwdArgs(_p(c_raw), st.w, st.y, None, _p(c_ss), None, ACT_RELU, st.N, st.Cin, st.Cout, st.T, st.H, st.W,
                               1, st.dtype, st.w_panel, in_add=_p(shortcut), in_add_scale_shift=_p(r_ss), in_store=_p(y))
            # (the fold must not cost a layer its stationary kernel: stages 4 / 5 fold where the weights-stationary kernel carries
            # the prologue itself -- x3d_pw_kernel_name of the folded form says which kernel it gets)
            if (fold_fwd and st.stride == 1 and pl.lib.x3d_pw_fwd_tail_supported(C.byref(ft))
                    and (not hip.pw_kernel_name(st).startswith(("pw_gemm_wst", "pw_gemm_ws_kernel"))
                         or (hip.pw_kernel_name(ft).startswith("pw_gemm_wst") and self.opt["tail_fold_wst"]))):
                if pl.blocks:
                    pl.blocks[-1].tail_fwd_folded = True
                else:
                    pl.stem_tail_folded = True      # the stem's BatchNorm + ReLU (no Add)
                return ft
            pl.rec(F, "x3d_tail_fwd", c_raw, c_ss, shortcut, r_ss, y, n, cout_, p_out_, dt)
            return st

        x_cur, hh, ww = pl.y0, h1, w1
        pl.blocks = []
        for b in a.blocks:
            pre = block_prefix(b)
            q = f"{pre}/bottleneck"
            ho, wo = same_pad(hh, 3, b.stride)[0], same_pad(ww, 3, b.stride)[0]
            P_in, P_out = t * hh * ww, t * ho * wo

            class B:
                pass

            B.spec, B.x, B.hh, B.ww, B.ho, B.wo = b, x_cur, hh, ww, ho, wo
            B.a_raw = pl.act(n, b.inner, t, hh, ww)
            B.b_raw = pl.act(n, b.inner, t, ho, wo)
            B.c_raw = pl.act(n, b.cout, t, ho, wo)
            B.y = pl.act(n, b.cout, t, ho, wo)
            B.bn_a, B.bn_b, B.bn_c = bn_bufs(f"{q}/bn_a", b.inner), bn_bufs(f"{q}/bn_b", b.inner), bn_bufs(f"{q}/bn_c", b.cout)
            B.pool = pl.acc64(n, b.inner) if b.has_se else None
            B.gate = pl.f32(n, b.inner) if b.has_se else None
            B.hidden = pl.f32(n, b.se_width) if b.has_se else None
            # a: 1x1x1 on the block input -- materialised and already activated, or (16-bit storage, resident-panel kernel) built
            # on load from the raw `c` output + shortcut of the block below, whose residual tail is then not a pass of its own
            sa = hip.PwFwdArgs(_p(x_cur), _p(p[f"{q}/a/kernel"]), _p(B.a_raw), None, None, None, ACT_NONE, n, b.cin,
                               b.inner, t, hh, ww, 1, dt)
            sa.w_panel = self._wp(f"{q}/a/kernel")
            sa = fold_pending_tail(sa)
            B.sa = sa
            pl.rec(F, "x3d_pw_fwd", ("stats", sa, B.bn_a.stats))
            # b: channelwise 3x3x3, BN_a (+ its finalize when folded) + ReLU folded into the load, BN_b statistics + SE
            # squeeze in the epilogue
            if fold_on:
                sb = hip.Dw3dFwdArgs(_p(B.a_raw), _p(p[f"{q}/b/kernel"]), _p(B.b_raw), None, ACT_RELU, None, None,
                                     n, b.inner, t, hh, ww, b.stride, dt, C.pointer(bn_fold(B.bn_a, n * P_in)))
            else:
                bn_finish(B.bn_a, n * P_in)
                sb = hip.Dw3dFwdArgs(_p(B.a_raw), _p(p[f"{q}/b/kernel"]), _p(B.b_raw), _p(B.bn_a.ss), ACT_RELU, None, None,
                                     n, b.inner, t, hh, ww, b.stride, dt)
            B.sb = sb
            pl.rec(F, "x3d_dw3d_fwd", ("dwstats", sb, B.bn_b.stats, B.pool))
            bn_finish(B.bn_b, n * P_out)
            if b.has_se:
                pl.rec(F, "x3d_se_fwd", ("acc", B.pool), float(P_out), B.bn_b.ss, p[f"{q}/se_fc1/kernel"],
                       p[f"{q}/se_fc1/bias"], p[f"{q}/se_fc2/kernel"], p[f"{q}/se_fc2/bias"], B.gate, B.hidden, n,
                       b.inner, b.se_width)
            # c: 1x1x1 with BN_b * SE gate -> swish folded into the load
            sc = hip.PwFwdArgs(_p(B.b_raw), _p(p[f"{q}/c/kernel"]), _p(B.c_raw), None, _p(B.bn_b.ss), _p(B.gate),
                               ACT_SWISH, n, b.inner, b.cout, t, ho, wo, 1, dt)
            sc.w_panel = self._wp(f"{q}/c/kernel")
            B.sc = sc
            pl.rec(F, "x3d_pw_fwd", ("stats", sc, B.bn_c.stats))
            if not fold_on:
                bn_finish(B.bn_c, n * P_out)
            if b.has_shortcut_conv:
                B.r_raw = pl.act(n, b.cout, t, ho, wo)
                B.bn_r = bn_bufs(f"{pre}/bn_r", b.cout)
                B.xs = None
                if b.stride == 2 and wo % 2 == 1 and self.opt["shortcut_compact"]:
                    # the pixels the strided conv samples (reference model.py:360-367), copied once per step: the conv's forward and
                    # both of its gradients are then dense launches -- 16-byte coalesced rows instead of one output per 4-byte load.
                    # Only where the gather is at its worst (odd output rows): measured per launch on X3D-M (profiles/
                    # r06_ab_shortcut_compact.txt) the copy costs what the dense launches save at 112 / 56 / 28-wide inputs
                    # (108 + 39 + 25 us against -111 / -43 / -32) and a quarter of it at 14 -> 7 (15 against -68)
                    B.xs = pl.act(n, b.cin, t, ho, wo)
                    pl.rec(F, "x3d_subsample2", x_cur, B.xs, n * b.cin * t, hh, ww, dt)
                    sr = hip.PwFwdArgs(_p(B.xs), _p(p[f"{pre}/residual/kernel"]), _p(B.r_raw), None, None, None, ACT_NONE,
                                       n, b.cin, b.cout, t, ho, wo, 1, dt)
                else:
                    sr = hip.PwFwdArgs(_p(x_cur), _p(p[f"{pre}/residual/kernel"]), _p(B.r_raw), None, None, None, ACT_NONE,
                                       n, b.cin, b.cout, t, hh, ww, b.stride, dt)
                sr.w_panel = self._wp(f"{pre}/residual/kernel")
                B.sr = sr
                pl.rec(F, "x3d_pw_fwd", ("stats", sr, B.bn_r.stats))
                if fold_on:
                    pl.rec(F, "x3d_tail_fwd_bn", B.c_raw, bn_fold(B.bn_c, n * P_out), B.r_raw, bn_fold(B.bn_r, n * P_out),
                           B.y, n, b.cout, P_out, dt)
                else:
                    bn_finish(B.bn_r, n * P_out)
                    pending["tail"] = (B.c_raw, B.bn_c.ss, B.r_raw, B.bn_r.ss, B.y, b.cout, P_out)
            else:
                B.r_raw, B.bn_r = None, None
                if fold_on:
                    pl.rec(F, "x3d_tail_fwd_bn", B.c_raw, bn_fold(B.bn_c, n * P_out), x_cur, None, B.y, n, b.cout, P_out, dt)
                else:
                    pending["tail"] = (B.c_raw, B.bn_c.ss, x_cur, None, B.y, b.cout, P_out)
            B.tail_fwd_folded = None          # set by the consumer that takes the tail
            pl.blocks.append(B)
            x_cur, hh, ww = B.y, ho, wo

        # ---- head ------------------------------------------------------------------------------
        c_last, c5 = a.stages[-1].cout, a.conv5_out
        P5 = t * hh * ww
        pl.P5, pl.h5, pl.w5 = P5, hh, ww
        pl.y_last = x_cur
        pl.c5_raw = pl.act(n, c5, t, hh, ww)
        pl.bn5 = bn_bufs("conv5/layer_with_weights-1", c5)
        s5 = hip.PwFwdArgs(_p(x_cur), _p(p["conv5/layer_with_weights-0/kernel"]), _p(pl.c5_raw), None, None, None,
                           ACT_NONE, n, c_last, c5, t, hh, ww, 1, dt)
        s5.w_panel = self._wp("conv5/layer_with_weights-0/kernel")
        s5 = fold_pending_tail(s5)
        pl.s5 = s5
        pl.rec(F, "x3d_pw_fwd", ("stats", s5, pl.bn5.stats))
        bn_finish(pl.bn5, n * P5)
        pl.pooled = pl.f32(n, c5)
        pl.h1 = pl.f32(n, a.fc1_out)
        pl.logits = pl.f32(n, a.num_classes)
        pl.probs = pl.f32(n, a.num_classes)
        pl.rec(F, "x3d_pool_fwd", pl.c5_raw, pl.bn5.ss, pl.pooled, n, c5, P5, dt)
        pl.rec(F, "x3d_dense_fwd", pl.pooled, None, 1.0, p["fc1/kernel"], None, pl.h1, ACT_RELU, n, c5, a.fc1_out)
        use_drop = training and a.dropout_rate > 0
        pl.drop_mask = pl.f32(n, a.fc1_out) if use_drop else None
        pl.drop_scale = 1.0 / (1.0 - a.dropout_rate) if use_drop else 1.0
        pl.rec(F, "x3d_dense_fwd", pl.h1, pl.drop_mask, float(pl.drop_scale), p["fc2/kernel"], p["fc2/bias"],
               pl.logits, ACT_NONE, n, a.fc1_out, a.num_classes)
        if training:
            pl.labels = torch.zeros(n, dtype=torch.int32, device=self.device)
            pl.loss_rows = pl.f32(n)
            pl.dlogits = pl.f32(n, a.num_classes)
            pl.grad_scale_slot = len(F)
            pl.rec(F, "x3d_softmax_xent", pl.logits, pl.labels, pl.probs, pl.loss_rows, pl.dlogits, 1.0 / n, n,
                   a.num_classes)
            self._record_backward(pl)
        else:
            pl.rec(F, "x3d_softmax_xent", pl.logits, None, pl.probs, None, None, 1.0, n, a.num_classes)
            if n % a.num_preds:
                raise ValueError(f"inference batch {n} is not a multiple of views*crops={a.num_preds} "
                                 "(reference model.py:125)")
            pl.out = pl.f32(n // a.num_preds, a.num_classes)
            pl.rec(F, "x3d_view_mean", pl.probs, pl.out, n // a.num_preds, a.num_preds, a.num_classes)

        if not training:
            items = (hip.BnEvalItem * len(pl.bn_eval_items))(*pl.bn_eval_items)
            pl.bn_eval_table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
            F[0] = ("x3d_bn_eval_coef_batched", pl.lib.x3d_bn_eval_coef_batched,
                    (pl.bn_eval_table.data_ptr(), len(pl.bn_eval_items), float(eps)))
        # resolve fp64 accumulator handles into pointers
        pl.finalize_acc()
        for f, handle in pl.folds:
            f.stats = pl._zero_views[handle].data_ptr()
        for f, handle in getattr(pl, "bwd_folds", []):
            f.sums = pl._zero_views[handle].data_ptr()
        self._resolve(pl, pl.fwd)
        self._resolve(pl, pl.bwd)
        pl.wrap_side(pl.bwd)
        return pl

    @staticmethod
    def _resolve(pl: _Plan, lst):
        """Replace ('acc', handle) / ('stats', struct, handle) placeholders with device pointers."""
        for i, (name, fn, args) in enumerate(lst):
            new = []
            for a_ in args:
                if isinstance(a_, tuple) and a_ and a_[0] == "acc":
                    new.append(None if a_[1] is None else pl._zero_views[a_[1]].data_ptr())
                elif isinstance(a_, tuple) and a_ and a_[0] == "stats":
                    st = a_[1]
                    st.stats = None if a_[2] is None else pl._zero_views[a_[2]].data_ptr()
                    pl.keep.append(st)
                    new.append(C.byref(st))
                elif isinstance(a_, tuple) and a_ and a_[0] == "dwstats":
                    st = a_[1]
                    st.stats = None if a_[2] is None else pl._zero_views[a_[2]].data_ptr()
                    st.pool = None if a_[3] is None else pl._zero_views[a_[3]].data_ptr()
                    pl.keep.append(st)
                    new.append(C.byref(st))
                elif isinstance(a_, tuple) and a_ and a_[0] == "field":
                    st = a_[1]
                    for fname, handle in a_[2].items():
                        setattr(st, fname, None if handle is None else pl._zero_views[handle].data_ptr())
                    pl.keep.append(st)
                    new.append(C.byref(st))
                else:
                    new.append(a_)
            lst[i] = (name, fn, tuple(new))

    # ---------------------------------------------------------------------------------------------
    # backward: written out explicitly (the reference relies on Keras autodiff)
    # ---------------------------------------------------------------------------------------------
    def _record_backward(self, pl: _Plan):
        a, p, g = self.arch, self.params, self.grads
        n, t = pl.n, pl.t
        dt = hip.dtype_code(self.dtype)
        Bk = pl.bwd
        c5 = a.conv5_out
        # scratch shared by all blocks (sized for the largest user)
        max_out = max([pl.y0.numel()] + [B.y.numel() for B in pl.blocks])
        max_inner_out = max(B.b_raw.numel() for B in pl.blocks)
        max_inner_in = max(B.a_raw.numel() for B in pl.blocks)
        max_r = max([1] + [n * B.spec.cin * t * B.ho * B.wo for B in pl.blocks if B.spec.has_shortcut_conv])
        max_nc = max(n * B.spec.inner for B in pl.blocks)
        flat = lambda numel: pl.act(numel)
        pl.gbuf = [flat(max_out), flat(max_out)]
        pl.dv = flat(max_inner_out)
        pl.ga = flat(max_inner_in)
        pl.rtmp = flat(max_r)
        pl.coef_nc = pl.f32(max_nc * 4)
        pl.se_scratch = pl.f32(max([1] + [n * (2 * B.spec.inner + B.spec.se_width) for B in pl.blocks if B.spec.has_se]))
        pl.g5 = pl.act(*pl.c5_raw.shape)
        pl.dh1 = pl.f32(n, a.fc1_out)
        pl.dpooled = pl.f32(n, c5)
        pl.ds = None if pl.stem_fused else pl.act(*pl.s_raw.shape)

        # ---- head ------------------------------------------------------------------------------
        pl.rec(Bk, "x3d_dense_bwd", pl.dlogits, None, ACT_NONE, pl.h1, pl.drop_mask, float(pl.drop_scale),
               p["fc2/kernel"], pl.dh1, g["fc2/kernel"], g["fc2/bias"], n, a.fc1_out, a.num_classes)
        pl.rec(Bk, "x3d_dense_bwd", pl.dh1, pl.h1, ACT_RELU, pl.pooled, None, 1.0, p["fc1/kernel"], pl.dpooled,
               g["fc1/kernel"], None, n, c5, a.fc1_out)
        b5 = pl.bn5
        pl.rec(Bk, "x3d_relu_bn_bwd_reduce", None, pl.dpooled, pl.c5_raw, b5.ss, pl.g5, ("acc", b5.bsums), n, c5,
               pl.P5, dt)
        # The BatchNorm-backward finalize (sums -> the coefficient table of dYraw = A g + B yraw + C, dgamma, dbeta) is a ~6 us launch
        # between the producer of the sums and the consumers of the table.  Where EVERY consumer's kernel takes `coef_fold`
        # (include/x3d_hip.h x3d_bn_bwd_fold: the persistent weights-stationary kernels and the 16-bit weight-gradient kernel
        # -- stages 4 / 5 and conv5, 37 launches of an X3D-M step) the consumers derive the table themselves, the same bits, and
        # one of them -- never the weight-gradient launch -- publishes dgamma / dbeta and the table; no launch is recorded.
        fold_coef = self.opt["coef_fold"] and not pl.side_on
        pl.bwd_folds = getattr(pl, "bwd_folds", [])

        def fold_bn_bwd(bn, count, gamma, dgamma, dbeta, consumers):
            """consumers: [(argument struct, "dgrad" | "wgrad" | "bwd")] -- every launch that reads bn.coef.  True: folded."""
            if not fold_coef or not consumers:
                return False
            slot = {"dgrad": 0, "wgrad": 1, "bwd": 2}
            for st, kind in consumers:
                q3 = [None, None, None]
                q3[slot[kind]] = C.byref(st)
                if not pl.lib.x3d_pw_coef_fold_supported(*q3):
                    return False
            pub = next((st for st, kind in consumers if kind != "wgrad"), None)
            if pub is None:
                return False
            for st, kind in consumers:
                f = hip.BnBwdFold(None, float(count), _p(bn.mi), _p(gamma), _p(dgamma) if st is pub else None,
                                  _p(dbeta) if st is pub else None, _p(bn.coef) if st is pub else None)
                pl.keep += [f, bn.mi, gamma, dgamma, dbeta, bn.coef]
                pl.bwd_folds.append((f, bn.bsums))      # (the sums pointer is resolved with the other fp64 accumulators)
                st.coef_fold = hip.fold_address(f)
            return True

        c_last = a.stages[-1].cout
        w5 = hip.PwWgradArgs(_p(pl.g5), _p(pl.c5_raw), _p(b5.coef), _p(pl.y_last), None, None, ACT_NONE,
                             _p(g["conv5/layer_with_weights-0/kernel"]), n, c_last, c5, t, pl.h5, pl.w5, 1, dt)
        cur = 0
        dy = pl.gbuf[cur][:pl.y_last.numel()]
        d5 = hip.PwDgradArgs(_p(pl.g5), _p(pl.c5_raw), _p(b5.coef), _p(p["conv5/layer_with_weights-0/kernel"]),
                             _p(dy), EPI_STORE, None, None, None, None, None, n, c_last, c5, t, pl.h5, pl.w5, dt)
        d5.w_panel = self._wp("conv5/layer_with_weights-0/kernel", True)
        if not fold_bn_bwd(b5, n * pl.P5, p[f"{b5.prefix}/gamma"], g[f"{b5.prefix}/gamma"], g[f"{b5.prefix}/beta"],
                           [(w5, "wgrad"), (d5, "dgrad")]):
            pl.rec(Bk, "x3d_bn_bwd_finalize", ("acc", b5.bsums), float(n * pl.P5), b5.mi, p[f"{b5.prefix}/gamma"],
                   b5.coef, g[f"{b5.prefix}/gamma"], g[f"{b5.prefix}/beta"], c5)
        pl.rec_side(Bk, "x3d_pw_wgrad", w5)
        pl.rec(Bk, "x3d_pw_dgrad", d5)
        pl.rec_join(Bk)
        pl.bwd_stage_marks[len(a.stages)] = len(Bk)   # head finished

        # ---- residual blocks, last to first ----------------------------------------------------
        # The Add + ReLU backward of a block (g = dy * [y > 0] with the BN_c / BN_r backward sums) is applied by the kernel that
        # PRODUCES dy -- the `a`-conv backward of the next block, whose conv input is this block's y -- wherever the fused
        # x3d_pw_bwd covers that layer with its tail epilogue; x3d_tail_bwd remains for the other blocks (and tail_bwd_fold = False)
        fold_tail = self._fuse_pw_bwd and self.opt["tail_bwd_fold"]
        # The per-step operands of the recomputed-output `a` backward ride on the BatchNorm-backward finalize launches that
        # are on the critical path anyway (x3d_bn_bwd_finalize_rc): the panel of a layer with ITS bn_a finalize, the dW of a
        # layer with the NEXT finalize recorded after its x3d_pw_bwd (pw_bwd_rc_merge = False: separate launches).
        merge_rc = self.opt["pw_bwd_rc_merge"]
        pending_fin = {"job": None}
        # Weight-gradient SLABS (x3d_hip.h dw_slab): the persistent fused backward kernels of stage 4 end in a flush of 256
        # workgroups x [Cout][Cin] floats -- as device-scope atomics 13-24 us of a 85-105 us launch (profiles/r05_noflush.txt), as
        # plain stores into a slab per workgroup a few.  The slabs are added up by extra workgroups of the NEXT x3d_se_bnb_bwd
        # launch (every block has one, 12 us of latency on the critical path anyway): the `c` conv's by its own block's, the `a`
        # conv's by the block below's.  Two slab buffers per role, reused by every block (stream order).
        slab_on = self.opt["dw_slab"]
        slab_bufs = {}
        pending_reduce = {"a": None}

        def dw_slab_job(st, role, dw):
            """st: the x3d_pw_bwd / x3d_pw_wgrad arguments about to be recorded; returns its reduce job (and points st at the
            slab) or None."""
            query = pl.lib.x3d_pw_wgrad_dw_parts if isinstance(st, hip.PwWgradArgs) else pl.lib.x3d_pw_bwd_dw_parts
            # (small weight gradients -- stages 2 / 3 of the unfused fp32 path -- flush a few MB: not worth a reduce job)
            parts = int(query(C.byref(st))) if (slab_on and st.Cout * st.Cin >= 8192) else 0
            if parts <= 0:
                return None
            elems = st.Cout * st.Cin
            buf = slab_bufs.get((role, parts * elems))
            if buf is None:
                buf = slab_bufs[(role, parts * elems)] = pl.f32(parts * elems)
                pl.keep.append(buf)
            st.dw_slab, st.dw_slab_parts = _p(buf), parts
            return hip.DwReduceJob(_p(buf), _p(dw), parts, elems)

        def flush_pending_reduce():
            """a slab nobody has added up yet, in front of a point where its gradient must be final: its own small launch"""
            job, pending_reduce["a"] = pending_reduce["a"], None
            if job is not None:
                jobs = (hip.DwReduceJob * 1)(job)
                pl.keep.append(jobs)
                pl.rec(Bk, "x3d_dw_slab_reduce", jobs, 1)

        pending_mark = {"stage": None}

        def rec_bn_bwd_finalize(bn, count, gamma, dgamma, dbeta, c, prep=None, consumers=None):
            if prep is None and pending_fin["job"] is None and fold_bn_bwd(bn, count, gamma, dgamma, dbeta, consumers):
                return          # (derived by the consumers: no launch)
            fin, pending_fin["job"] = pending_fin["job"], None
            if prep is None and fin is None:
                pl.rec(Bk, "x3d_bn_bwd_finalize", ("acc", bn.bsums), float(count), bn.mi, gamma, bn.coef, dgamma, dbeta, c)
                return
            w_, panel_, c0_, cin_ = prep if prep is not None else (None, None, None, 0)
            f_ = fin if fin is not None else (None, None, None, None, 0, 0)
            pl.rec(Bk, "x3d_bn_bwd_finalize_rc", ("acc", bn.bsums), float(count), bn.mi, gamma, bn.coef, dgamma, dbeta, c,
                   w_, panel_, c0_, cin_, ("acc", f_[0]) if f_[0] is not None else None, f_[1], f_[2], f_[3], f_[4], f_[5], dt)
            if fin is not None and pending_mark["stage"] is not None:
                # this launch finished the dW of the FIRST block of a stage (its `a` conv's pending job): only now is every
                # gradient of that stage final -- the stage's all-reduce bucket may start behind it, not before
                pl.bwd_stage_marks[pending_mark["stage"]], pending_mark["stage"] = len(Bk), None

        for bi in range(len(pl.blocks) - 1, -1, -1):
            B = pl.blocks[bi]
            prev = pl.blocks[bi - 1] if bi > 0 else None
            b: BlockSpec = B.spec
            pre = block_prefix(b)
            q = f"{pre}/bottleneck"
            P_in, P_out = t * B.hh * B.ww, t * B.ho * B.wo
            B.bwd_start, B.dy_view = len(Bk), dy.view(B.y.shape)
            B.tail_folded = getattr(B, "tail_folded", False)
            if not B.tail_folded:
                # dy -> g = dy*[y>0] in place, with the BN_c (and BN_r) backward sums
                pl.rec(Bk, "x3d_tail_bwd", dy, B.y, B.c_raw, B.r_raw, ("acc", B.bn_c.bsums),
                       ("acc", B.bn_r.bsums) if B.bn_r else None, n, b.cout, P_out, dt)
            gten = dy
            # c
            wc = hip.PwWgradArgs(_p(gten), _p(B.c_raw), _p(B.bn_c.coef), _p(B.b_raw), _p(B.bn_b.ss), _p(B.gate),
                                 ACT_SWISH, _p(g[f"{q}/c/kernel"]), n, b.inner, b.cout, t, B.ho, B.wo, 1, dt)
            B.nc_sums = pl.acc64(n, b.inner, 2)
            dvv = pl.dv[:B.b_raw.numel()]
            dc = hip.PwDgradArgs(_p(gten), _p(B.c_raw), _p(B.bn_c.coef), _p(p[f"{q}/c/kernel"]), _p(dvv),
                                 EPI_SWISH_BWD, None, _p(B.b_raw), _p(B.bn_b.ss), _p(B.gate), None, n, b.inner,
                                 b.cout, t, B.ho, B.wo, dt)
            dc.w_panel = self._wp(f"{q}/c/kernel", True)
            # one pass over g / c_raw / b_raw for both gradients where the fused kernel covers the layer
            fc = hip.PwBwdArgs(_p(gten), _p(B.c_raw), _p(B.bn_c.coef), dc.w_panel, _p(dvv), EPI_SWISH_BWD, None,
                               _p(B.b_raw), _p(B.bn_b.ss), _p(B.gate), None, None, _p(g[f"{q}/c/kernel"]), n, b.inner,
                               b.cout, t, B.ho, B.wo, dt)
            c_job = None
            c_fused = bool(self._fuse_pw_bwd and pl.lib.x3d_pw_bwd_supported(C.byref(fc)))
            rec_bn_bwd_finalize(B.bn_c, n * P_out, p[f"{q}/bn_c/gamma"], g[f"{q}/bn_c/gamma"], g[f"{q}/bn_c/beta"], b.cout,
                                consumers=[(fc, "bwd")] if c_fused else [(wc, "wgrad"), (dc, "dgrad")])
            if c_fused:
                c_job = dw_slab_job(fc, "c", g[f"{q}/c/kernel"])
                pl.rec(Bk, "x3d_pw_bwd", ("field", fc, {"nc_sums": B.nc_sums}))
            else:
                # Weight gradients feed nothing but the optimizer: they run on the side stream next to the data-gradient
                # chain.  What they read (g, the raw conv outputs, the shared `ga` scratch) is next overwritten by the
                # following depthwise backward / the block after it, and every depthwise backward is preceded by a join.
                if not pl.side_on:
                    c_job = dw_slab_job(wc, "c", g[f"{q}/c/kernel"])
                pl.rec_side(Bk, "x3d_pw_wgrad", wc)
                pl.rec(Bk, "x3d_pw_dgrad", ("field", dc, {"nc_sums": B.nc_sums}))
            # SE + BN_b backward from the per-(n,c) sums
            se = hip.SeBnbBwdArgs(
                None, None, float(P_out), _p(B.bn_b.ss), _p(B.bn_b.mi), _p(p[f"{q}/bn_b/gamma"]),
                _p(p.get(f"{q}/se_fc1/kernel")), _p(p.get(f"{q}/se_fc1/bias")), _p(p.get(f"{q}/se_fc2/kernel")),
                _p(p.get(f"{q}/se_fc2/bias")), _p(B.gate), _p(B.hidden), _p(g.get(f"{q}/se_fc1/kernel")),
                _p(g.get(f"{q}/se_fc1/bias")), _p(g.get(f"{q}/se_fc2/kernel")), _p(g.get(f"{q}/se_fc2/bias")),
                _p(g[f"{q}/bn_b/gamma"]), _p(g[f"{q}/bn_b/beta"]), _p(pl.coef_nc), _p(pl.se_scratch), n, b.inner,
                b.se_width)
            if c_job is not None:
                se.reduce[0] = c_job
            if pending_reduce["a"] is not None:      # the `a` conv of the block above (recorded just before this block)
                se.reduce[1], pending_reduce["a"] = pending_reduce["a"], None
            pl.rec(Bk, "x3d_se_bnb_bwd", ("field", se, {"nc_sums": B.nc_sums, "pool_sums": B.pool}))
            # b (fused data + weight gradient), emits grad wrt BN_a output with the ReLU mask applied
            gaa = pl.ga[:B.a_raw.numel()]
            db = hip.Dw3dBwdArgs(_p(dvv), _p(B.b_raw), _p(pl.coef_nc), _p(B.a_raw), _p(B.bn_a.ss),
                                 _p(p[f"{q}/b/kernel"]), _p(gaa), None, _p(g[f"{q}/b/kernel"]), n, b.inner, t, B.hh,
                                 B.ww, b.stride, dt)
            B.db = db
            pl.rec_join(Bk)
            pl.rec(Bk, "x3d_dw3d_bwd", ("field", db, {"a_sums": B.bn_a.bsums}))
            # (bn_a's backward finalize is recorded below, right in front of the `a` backward: whether it also builds that launch's
            # panel is known there; the shortcut launches in between do not depend on it)
            # a
            wa = hip.PwWgradArgs(_p(gaa), _p(B.a_raw), _p(B.bn_a.coef), _p(B.x), None, None, ACT_NONE,
                                 _p(g[f"{q}/a/kernel"]), n, b.cin, b.inner, t, B.hh, B.ww, 1, dt)
            nxt = pl.gbuf[1 - cur][:B.x.numel()]
            if b.has_shortcut_conv:
                rt = pl.rtmp[:n * b.cin * P_out]
                # the strided shortcut conv's two gradients in ONE launch over g and the even pixels of the block input, its raw
                # output recomputed algebraically like the `a` conv's (pw_bwd_rc.hip, x_stride = 2) -- where the shape is covered
                sr = None
                per = int(pl.lib.x3d_pw_bwd_rc_panel_elems(b.cout, b.cin)) if (self._fuse_pw_bwd and self._rc_pw_bwd and b.stride == 2
                                                                                and self.dtype != torch.float32) else 0
                xs = getattr(B, "xs", None)       # the even-pixel copy of B.x the forward pass made (option shortcut_compact)
                if per:
                    rcr = (pl.act(per), pl.f32(b.cin), pl.acc64((int(pl.lib.x3d_pw_bwd_rc_sums_elems(b.cout, b.cin)) + 1) // 2))
                    if xs is not None:
                        sr = hip.PwBwdArgs(_p(gten), None, None, None, _p(rt), EPI_STORE, None, None, None, None, None, _p(xs), None,
                                           n, b.cin, b.cout, t, B.ho, B.wo, dt, None, None, None, None, _p(rcr[0]), _p(rcr[1]), None)
                    else:
                        sr = hip.PwBwdArgs(_p(gten), None, None, None, _p(rt), EPI_STORE, None, None, None, None, None, _p(B.x), None,
                                           n, b.cin, b.cout, t, B.ho, B.wo, dt, None, None, None, None, _p(rcr[0]), _p(rcr[1]), None,
                                           b.stride, B.hh, B.ww)
                    if not pl.lib.x3d_pw_bwd_supported(C.byref(sr)):
                        sr = None
                B.r_bwd_rc = sr is not None
                w_r, g_r = p[f"{pre}/residual/kernel"], g[f"{pre}/residual/kernel"]
                if sr is not None:
                    if merge_rc:
                        rec_bn_bwd_finalize(B.bn_r, n * P_out, p[f"{pre}/bn_r/gamma"], g[f"{pre}/bn_r/gamma"], g[f"{pre}/bn_r/beta"],
                                            b.cout, prep=(w_r, rcr[0], rcr[1], b.cin))
                    else:
                        rec_bn_bwd_finalize(B.bn_r, n * P_out, p[f"{pre}/bn_r/gamma"], g[f"{pre}/bn_r/gamma"], g[f"{pre}/bn_r/beta"], b.cout)
                        pl.rec(Bk, "x3d_pw_bwd_rc_prepare", w_r, B.bn_r.coef, rcr[0], rcr[1], b.cout, b.cin, dt)
                    pl.rec(Bk, "x3d_pw_bwd", ("field", sr, {"rc_sums": rcr[2]}))
                    if merge_rc:     # (the bn_a finalize recorded next carries this dW)
                        pending_fin["job"] = (rcr[2], w_r, B.bn_r.coef, g_r, b.cout, b.cin)
                    else:
                        pl.rec(Bk, "x3d_pw_bwd_rc_finish", ("acc", rcr[2]), w_r, B.bn_r.coef, g_r, b.cout, b.cin, dt)
                else:
                    if xs is not None:
                        wr = hip.PwWgradArgs(_p(gten), _p(B.r_raw), _p(B.bn_r.coef), _p(xs), None, None, ACT_NONE,
                                             _p(g_r), n, b.cin, b.cout, t, B.ho, B.wo, 1, dt)
                    else:
                        wr = hip.PwWgradArgs(_p(gten), _p(B.r_raw), _p(B.bn_r.coef), _p(B.x), None, None, ACT_NONE,
                                             _p(g_r), n, b.cin, b.cout, t, B.hh, B.ww, b.stride, dt)
                    dr = hip.PwDgradArgs(_p(gten), _p(B.r_raw), _p(B.bn_r.coef), _p(w_r), _p(rt),
                                         EPI_STORE, None, None, None, None, None, n, b.cin, b.cout, t, B.ho, B.wo, dt)
                    dr.w_panel = self._wp(f"{pre}/residual/kernel", True)
                    rec_bn_bwd_finalize(B.bn_r, n * P_out, p[f"{pre}/bn_r/gamma"], g[f"{pre}/bn_r/gamma"], g[f"{pre}/bn_r/beta"], b.cout,
                                        consumers=[(wr, "wgrad"), (dr, "dgrad")])
                    pl.rec_side(Bk, "x3d_pw_wgrad", wr)
                    pl.rec(Bk, "x3d_pw_dgrad", dr)
                da = hip.PwDgradArgs(_p(gaa), _p(B.a_raw), _p(B.bn_a.coef), _p(p[f"{q}/a/kernel"]), _p(nxt),
                                     EPI_ADD_STRIDED if b.stride == 2 else EPI_ADD, _p(rt), None, None, None, None, n,
                                     b.cin, b.inner, t, B.hh, B.ww, dt)
            else:
                da = hip.PwDgradArgs(_p(gaa), _p(B.a_raw), _p(B.bn_a.coef), _p(p[f"{q}/a/kernel"]), _p(nxt), EPI_ADD,
                                     _p(gten), None, None, None, None, n, b.cin, b.inner, t, B.hh, B.ww, dt)
            da.w_panel = self._wp(f"{q}/a/kernel", True)
            # The `a` conv's raw output is linear in its input, so the BatchNorm backward dY = A g + B a_raw + C folds into the
            # GEMMs (pw_bwd_rc.hip): where that form covers the layer (stages 2-3 first blocks: Cin <= 32) the launch streams
            # g and x only -- a_raw, 2.25x the size of x, is not read here.  Per-step operands: the panel [W^T A | W^T B W] and
            # c0 (x3d_pw_bwd_rc_prepare, after bn_a's backward finalize) and the moment sums dW is finished from.
            rc = None
            pe = int(pl.lib.x3d_pw_bwd_rc_panel_elems(b.inner, b.cin)) if (self._fuse_pw_bwd and self._rc_pw_bwd and
                                                                            self.dtype != torch.float32) else 0
            if b.inner > 127 and not self.opt["pw_bwd_rc_wide"]:      # (the 48 -> 216 layer unfused as before)
                pe = 0
            if pe:
                rc = (pl.act(pe), pl.f32(b.cin), pl.acc64((int(pl.lib.x3d_pw_bwd_rc_sums_elems(b.inner, b.cin)) + 1) // 2))

            def a_bwd_args(tail_c=None, tail_r=None, use_rc=True):
                if rc is not None and use_rc:
                    return hip.PwBwdArgs(da.g, None, None, None, da.dx, da.epi, da.add, None, None, None, None, _p(B.x), None,
                                         n, b.cin, b.inner, t, B.hh, B.ww, dt, tail_c, tail_r, None, None, _p(rc[0]), _p(rc[1]), None)
                return hip.PwBwdArgs(da.g, da.yraw, da.coef, da.w_panel, da.dx, da.epi, da.add, None, None, None, None,
                                     _p(B.x), _p(g[f"{q}/a/kernel"]), n, b.cin, b.inner, t, B.hh, B.ww, dt, tail_c, tail_r, None, None)

            def supported(st):
                return st is not None and bool(pl.lib.x3d_pw_bwd_supported(C.byref(st)))

            fa = a_bwd_args()
            if rc is not None and not supported(fa):
                rc, fa = None, a_bwd_args(use_rc=False)
            ft = None
            if fold_tail and prev is not None:   # B.x is prev.y: this launch can apply prev's Add + ReLU backward to its dx
                ft = a_bwd_args(_p(prev.c_raw), _p(prev.r_raw))
                if not supported(ft):
                    ft = None
            stem_ft = None
            if fold_tail and prev is None and self.opt["stem_bwd_fold"]:
                # B.x is the stem output y0 = relu(bn(t_raw)): the same epilogue masks dx with [y0 > 0] and takes the stem
                # BatchNorm's backward sums (sum dx, sum dx * t_raw) -- the x3d_relu_bn_bwd_reduce pass over dy0 / t_raw goes
                stem_ft = a_bwd_args(_p(pl.t_raw), None)
                if not supported(stem_ft):
                    stem_ft = None
            pl.stem_bwd_folded = stem_ft is not None
            fields = {} if rc is None else {"rc_sums": rc[2]}
            chosen = ft if ft is not None else (stem_ft if stem_ft is not None else (fa if self._fuse_pw_bwd and supported(fa) else None))
            B.a_bwd_rc = rc is not None and chosen is not None
            if B.a_bwd_rc and merge_rc:
                rec_bn_bwd_finalize(B.bn_a, n * P_in, p[f"{q}/bn_a/gamma"], g[f"{q}/bn_a/gamma"], g[f"{q}/bn_a/beta"], b.inner,
                                    prep=(p[f"{q}/a/kernel"], rc[0], rc[1], b.cin))
            else:
                rec_bn_bwd_finalize(B.bn_a, n * P_in, p[f"{q}/bn_a/gamma"], g[f"{q}/bn_a/gamma"], g[f"{q}/bn_a/beta"], b.inner,
                                    consumers=None if B.a_bwd_rc else ([(chosen, "bwd")] if chosen is not None else
                                                                       [(wa, "wgrad"), (da, "dgrad")]))
            if B.a_bwd_rc and not merge_rc:
                pl.rec(Bk, "x3d_pw_bwd_rc_prepare", p[f"{q}/a/kernel"], B.bn_a.coef, rc[0], rc[1], b.inner, b.cin, dt)
            # (the `a` conv's slab is added up by the NEXT block's x3d_se_bnb_bwd: not for the first block of a stage, whose
            # gradient must be final at the stage mark -- a reduce launch of its own would cost what the slab saves)
            if chosen is not None and rc is None and b.index != 0:
                pending_reduce["a"] = dw_slab_job(chosen, "a", g[f"{q}/a/kernel"])
            if ft is not None:
                prev.tail_folded = True
                pl.rec(Bk, "x3d_pw_bwd", ("field", ft, dict(fields, tail_sums_c=prev.bn_c.bsums,
                                                            tail_sums_r=prev.bn_r.bsums if prev.bn_r else None)))
            elif stem_ft is not None:
                pl.rec(Bk, "x3d_pw_bwd", ("field", stem_ft, dict(fields, tail_sums_c=pl.bn1.bsums)))
            elif chosen is not None:
                pl.rec(Bk, "x3d_pw_bwd", ("field", fa, fields))
            else:
                if not pl.side_on and b.index != 0:
                    pending_reduce["a"] = dw_slab_job(wa, "a", g[f"{q}/a/kernel"])
                pl.rec_side(Bk, "x3d_pw_wgrad", wa)
                pl.rec(Bk, "x3d_pw_dgrad", da)
            if B.a_bwd_rc:
                if merge_rc:    # dW rides on the next BatchNorm-backward finalize (the block below's bn_c, or the stem's)
                    pending_fin["job"] = (rc[2], p[f"{q}/a/kernel"], B.bn_a.coef, g[f"{q}/a/kernel"], b.inner, b.cin)
                else:
                    pl.rec(Bk, "x3d_pw_bwd_rc_finish", ("acc", rc[2]), p[f"{q}/a/kernel"], B.bn_a.coef, g[f"{q}/a/kernel"],
                           b.inner, b.cin, dt)
            cur = 1 - cur
            B.bwd_stop, B.dx_view = len(Bk), nxt.view(B.x.shape)
            dy = nxt
            if b.index == 0:
                flush_pending_reduce()                  # (the next x3d_se_bnb_bwd belongs to the stage below: behind this stage's mark)
                pl.rec_join(Bk)
                if pending_fin["job"] is not None:      # the dW of this block's `a` conv rides on the NEXT finalize launch:
                    pending_mark["stage"] = b.stage     # the mark is set there (rec_bn_bwd_finalize)
                else:
                    pl.bwd_stage_marks[b.stage] = len(Bk)   # every gradient of stages >= b.stage is final

        # ---- stem ------------------------------------------------------------------------------
        b1 = pl.bn1
        P1 = t * pl.y0.shape[3] * pl.y0.shape[4]
        # sums only (g = NULL): x3d_dwt_bwd applies the ReLU mask itself on the t_raw values it loads anyway, so the masked
        # gradient of the widest tensor of the network is neither written nor read back
        if not getattr(pl, "stem_bwd_folded", False):
            pl.rec(Bk, "x3d_relu_bn_bwd_reduce", dy, None, pl.t_raw, b1.ss, None, ("acc", b1.bsums), n, a.c1, P1, dt)
        rec_bn_bwd_finalize(b1, n * P1, p["conv1/bn/gamma"], g["conv1/bn/gamma"], g["conv1/bn/beta"], a.c1)
        assert pending_fin["job"] is None and pending_mark["stage"] is None and pending_reduce["a"] is None
        if pl.stem_fused:
            # one pass over dy, t_raw and the batch: conv_s recomputed on the matrix cores, the conv_t input gradient kept in LDS
            pl.input_slots.append((Bk, len(Bk), 4))
            pl.rec(Bk, "x3d_stem_bwd", dy, pl.t_raw, b1.ss, b1.coef, pl.x, p["conv1/conv_s/kernel"], p["conv1/conv_t/kernel"],
                   g["conv1/conv_s/kernel"], g["conv1/conv_t/kernel"], n, self.in_channels, t, pl.h, pl.w, a.c1,
                   a.c1_temp_filter, dt, 1)
        else:
            pl.rec(Bk, "x3d_dwt_bwd", dy, pl.t_raw, b1.ss, b1.coef, pl.s_raw, p["conv1/conv_t/kernel"], pl.ds,
                   g["conv1/conv_t/kernel"], n, a.c1, t, pl.y0.shape[3] * pl.y0.shape[4], a.c1_temp_filter, dt)
            pl.input_slots.append((Bk, len(Bk)))
            pl.rec(Bk, "x3d_stem_s_wgrad", pl.x, pl.ds, g["conv1/conv_s/kernel"], n, self.in_channels, t, pl.h, pl.w,
                   a.c1, dt, int(pl.x_cl))
        pl.rec_join(Bk)
        pl.bwd_stage_marks[-1] = len(Bk)

    # ---------------------------------------------------------------------------------------------
    # execution
    # ---------------------------------------------------------------------------------------------
    def _stem_fused(self, pl: _Plan, n, t, h, w, dt):
        """Does this plan run the stem as x3d_stem_fwd / x3d_stem_bwd (one launch each way)?  Needs the in-place channels-last
        input (pl.x_cl) and a shape the fused kernels take (x3d_stem_fused_supported: bit 0 forward, bit 1 backward -- a plan
        with a backward pass needs both, since the fused forward stores no conv_s output); the two-kernel path otherwise."""
        if not (pl.x_cl and self.opt["stem_fused"]):
            return False
        need = 3 if pl.training else 1
        have = pl.lib.x3d_stem_fused_supported(self.in_channels, self.arch.c1, self.arch.c1_temp_filter, n, t, h, w, dt, 1)
        return (have & need) == need

    def _bind_input(self, pl: _Plan, x):
        if self.dry:
            raise hip.X3DHipError("a dry model cannot run (no CPU fallback for the hot path)")
        if x.dim() != 5 or x.shape[-1] != self.in_channels:
            raise ValueError(f"expected a channels-last clip batch [N, T, H, W, {self.in_channels}], got {tuple(x.shape)}")
        if not x.is_cuda:
            x = x.to(self.device, non_blocking=True)
        x = x.contiguous()
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16) or (x.dtype != torch.float32 and x.dtype != self.dtype):
            x = x.float()
        n, t, h, w, c = x.shape
        if pl.x_cl:
            # the stem reads the batch where it lies: storage type of the model, 16-byte aligned (a copy only if it is neither)
            if x.dtype != self.dtype:
                x = x.to(self.dtype)
            if x.data_ptr() % 16:
                x = x.clone()
            for lst, i, *pos in pl.input_slots:
                k = pos[0] if pos else 0                   # (the batch is argument 0 of most stem launches, argument 4 of x3d_stem_bwd)
                name, fn, args = lst[i]
                lst[i] = (name, fn, tuple(args[:k]) + (x.data_ptr(),) + tuple(args[k + 1:]))
        else:
            hip.call("x3d_nthwc_to_ncthw", x.data_ptr(), hip.dtype_code(x.dtype), pl.x.data_ptr(),
                     hip.dtype_code(self.dtype), n, c, t * h * w)
        pl._x_keepalive = x      # (until the next batch is bound: the backward pass reads it again)

    def _draw_dropout(self, pl: _Plan):
        if pl.drop_mask is None:
            return
        if self._dropout_mask_override is not None:
            pl.drop_mask.copy_(self._dropout_mask_override.to(self.device, torch.float32))
        else:
            pl.drop_mask.bernoulli_(1.0 - self.arch.dropout_rate)

    def set_dropout_mask(self, mask: Optional[torch.Tensor]):
        """Fix the dropout keep-mask ([N, 2048] of 0/1) for reproducible parity tests; None = random."""
        self._dropout_mask_override = mask

    def __call__(self, input, training=False):
        return self.call(input, training)

    def call(self, input, training=False):
        """Forward pass (reference model.py:113-127).  Returns fp32 probabilities: ``[N, classes]`` when
        training, ``[N / (views*crops), classes]`` (view-averaged) otherwise."""
        n, t, h, w, _ = input.shape
        pl = self._plan(n, t, h, w, training)
        self._bind_input(pl, input)
        self._pack_panels()
        if training:
            pl.zero_buf.zero_()
            self._draw_dropout(pl)
            # forward only: stop before the loss kernel (labels unknown); softmax without labels
            pl.run(pl.fwd, 0, pl.grad_scale_slot)
            hip.call("x3d_softmax_xent", pl.logits.data_ptr(), None, pl.probs.data_ptr(), None, None, 1.0, n,
                     self.num_classes)
            return pl.probs
        pl.zero_buf.zero_()
        pl.run(pl.fwd)
        return pl.out

    def forward_backward(self, input, labels, global_batch=None, on_stage_done=None, loss_scale=1.0):
        """One training forward + backward.  Fills ``self.grads`` (data gradients only; the L2 term is
        applied by the optimizer), updates BN moving statistics, returns the plan (loss_rows, probs).

        global_batch: divisor of the loss mean (defaults to the local batch; data-parallel callers pass
            world_size * local batch so that summing gradients over ranks gives the global mean).
        loss_scale: every gradient is multiplied by this factor (Keras LossScaleOptimizer, reference train.py:99-100:
            keeps fp16 activation gradients out of the denormal range); the optimizer step divides it out again
            (`apply_sgd(grad_scale=1 / loss_scale)`).  The backward kernels are linear in the upstream gradient.
        on_stage_done(stage): called with "fwd" once the forward pass is on the stream, then as soon as every
            gradient of ``stage`` (4 = head, 3..0 = stages, -1 = stem) is final on the stream -- the hook gradient
            all-reduce buckets attach to.
        """
        n, t, h, w, _ = input.shape
        pl = self._plan(n, t, h, w, True)
        self._bind_input(pl, input)
        if not labels.is_cuda:   # host labels are validated for free; device labels by the kernel (NaN loss row, zero gradient)
            if labels.numel() != n or int(labels.min()) < 0 or int(labels.max()) >= self.num_classes:
                raise ValueError(f"labels must be {n} class indices in [0, {self.num_classes})")
        pl.labels.copy_(labels.to(self.device, non_blocking=True).to(torch.int32))
        # (round 4: the panel packing, the gradient-buffer zeroing and the dropout draw -- ~70 us the stem does not depend on -- on a
        # second stream beside the stem's two convolutions, joined in front of the first pointwise conv: 22.30 -> 22.63 ms per step,
        # three alternating runs on one box; the fork / join costs more than it hides.  Not kept.)
        self._pack_panels()
        pl.zero_buf.zero_()
        self.flat_grads.zero_()
        self._draw_dropout(pl)
        gb = float(global_batch or n)
        pl.run(pl.fwd, 0, pl.grad_scale_slot)
        hip.call("x3d_softmax_xent", pl.logits.data_ptr(), pl.labels.data_ptr(), pl.probs.data_ptr(),
                 pl.loss_rows.data_ptr(), pl.dlogits.data_ptr(), float(loss_scale) / gb, n, self.num_classes)
        if on_stage_done is None:
            pl.run(pl.bwd)
        else:
            on_stage_done("fwd")      # forward (and the BN moving-statistics updates in it) is on the stream
            marks = sorted(pl.bwd_stage_marks.items(), key=lambda kv: kv[1])
            start = 0
            for stage, stop in marks:
                pl.run(pl.bwd, start, stop)
                on_stage_done(stage)
                start = stop
        return pl

    def regularization_loss(self):
        """weight_decay * sum(w^2) over the L2-regularised kernels (reference model.py:47)."""
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        hip.call("x3d_l2_sumsq", self.flat_params.data_ptr(), self.l2_mask.data_ptr(), acc.data_ptr(),
                 self.n_trainable_flat)
        return acc * self.arch.weight_decay

    def apply_sgd(self, lr, momentum=0.9, grad_scale=1.0):
        """SGD(momentum, nesterov=True) + L2 (reference train.py:89-92, model.py:47), one launch."""
        self._claim_slots("sgd")
        hip.call("x3d_sgd_nesterov", self.flat_params.data_ptr(), self.flat_velocity.data_ptr(),
                 self.flat_grads.data_ptr(), self.l2_mask.data_ptr(), float(lr), float(momentum),
                 float(self.arch.weight_decay), float(grad_scale), self.n_trainable_flat)

    def apply_adam(self, lr, step, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """Adam + L2 (reference train.py:93-95: tf.optimizers.Adam(learning_rate), Keras defaults), one launch.  The first
        moment lives in `flat_velocity` (the slot the SGD branch uses for momentum), the second in `flat_second`."""
        self._claim_slots("adam")
        if getattr(self, "flat_second", None) is None:
            self.flat_second = torch.zeros_like(self.flat_velocity)
        hip.call("x3d_adam", self.flat_params.data_ptr(), self.flat_velocity.data_ptr(), self.flat_second.data_ptr(),
                 self.flat_grads.data_ptr(), self.l2_mask.data_ptr(), float(lr), float(beta1), float(beta2), float(eps),
                 float(self.arch.weight_decay), float(grad_scale), int(step), self.n_trainable_flat)

    def grads_finite(self) -> bool:
        """True when every entry of the flat gradient buffer is finite (x3d_all_finite; synchronises)."""
        if getattr(self, "_finite_flag", None) is None:
            self._finite_flag = torch.ones(1, dtype=torch.int32, device=self.device)
        self._finite_flag.fill_(1)
        hip.call("x3d_all_finite", self.flat_grads.data_ptr(), self.n_trainable_flat, self._finite_flag.data_ptr())
        return bool(self._finite_flag.item())

    def grad_bucket(self, stage):
        """Contiguous slice of the flat gradient buffer holding the gradients of one stage
        (4 = conv5 + head, 0..3 = residual stages, -1 = stem)."""
        names = [k for k in self.param_order if k in self.grads]
        if stage == -1:
            sel = [k for k in names if k.startswith("conv1/")]
        elif stage == len(self.arch.stages):
            sel = [k for k in names if k.startswith(("conv5/", "fc1/", "fc2/"))]
        else:
            sel = [k for k in names if k.startswith(f"stages/{stage}/")]
        lo = self._offsets[sel[0]]
        last = sel[-1]
        hi = self._offsets[last] + (self.params[last].numel() + 3) // 4 * 4
        return self.flat_grads[lo:hi]

    def moving_stats_flat(self):
        return self.flat_params[self.n_trainable_flat:]
