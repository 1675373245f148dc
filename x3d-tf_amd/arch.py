"""Architecture arithmetic for X3D: channel/depth rounding and the per-layer table.

Follows reference utils.py:7-40 (round_width / round_repeats) and model.py:31-76 (stage
construction).  Everything downstream (oracle, HIP module, checkpoint mapping, byte/FLOP
accounting for the roofline) is derived from the ``Arch`` built here.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple


def round_width(width, multiplier, min_depth=8, divisor=8):
    """Scale a channel count and snap it to a multiple of ``divisor`` (reference utils.py:7-30).

    Falsy multiplier returns ``width`` untouched; otherwise round-half-up to the divisor grid,
    floor at ``min_depth``, and step one divisor up when rounding lost more than 10 %.
    """
    if not multiplier:
        return width
    scaled = width * multiplier
    floor_ = min_depth or divisor
    snapped = max(floor_, int(scaled + divisor / 2) // divisor * divisor)
    if snapped < 0.9 * scaled:
        snapped += divisor
    return int(snapped)


def round_repeats(repeats, multiplier):
    """Scale a block count, rounding up (reference utils.py:32-40)."""
    if not multiplier:
        return repeats
    return int(math.ceil(multiplier * repeats))


def same_pad(in_size, kernel, stride):
    """TF 'SAME' padding rule -> (out_size, pad_before, pad_after).  [TF-3p rule, SURVEY Q3]"""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + kernel - in_size, 0)
    before = total // 2
    return out, before, total - before


@dataclass
class BlockSpec:
    stage: int              # 0..3  (res_stage_2..5)
    index: int              # position inside the stage
    global_index: int       # 1-based counter across the whole model (reference ResBlock._block_index)
    cin: int
    inner: int
    cout: int
    stride: int
    has_se: bool
    se_width: int
    has_shortcut_conv: bool


@dataclass
class StageSpec:
    cin: int
    inner: int
    cout: int
    depth: int
    blocks: List[BlockSpec] = field(default_factory=list)


@dataclass
class Arch:
    num_classes: int
    c1: int                      # stem output channels
    c1_temp_filter: int
    stages: List[StageSpec]
    conv5_out: int               # = inner width of the last stage (reference model.py:81)
    fc1_out: int                 # 2048
    bn_eps: float
    bn_momentum: float
    dropout_rate: float
    weight_decay: float
    num_preds: int               # TEST.NUM_TEMPORAL_VIEWS * TEST.NUM_SPATIAL_CROPS (model.py:25)
    se_ratio: float = 0.0625

    @property
    def blocks(self) -> List[BlockSpec]:
        return [b for s in self.stages for b in s.blocks]

    def out_shape(self, t, h, w) -> Tuple[int, int, int]:
        """(T, H, W) at the end of the last stage for an input clip of (t, h, w)."""
        h = (h + 2 - 3) // 2 + 1
        w = (w + 2 - 3) // 2 + 1
        for s in self.stages:
            h = same_pad(h, 3, 2)[0]
            w = same_pad(w, 3, 2)[0]
        return t, h, w


def build_arch(cfg) -> Arch:
    """cfg -> Arch, restating reference model.py:21-76.

    SE placement: the reference passes a *class-level* counter, already incremented,
    as ``block_index`` (model.py:350-351,378) and enables SE when ``(block_index + 1) % 2 == 0``
    (model.py:275,311).  In a fresh process that is global blocks 1,3,5,... counted across stages.
    Here the counter is explicit and per model, starting at 1.
    """
    net = cfg.NETWORK
    if net.SCALE_RES2:
        c1 = round_width(net.C1_CHANNELS, net.WIDTH_FACTOR)
        mult = 1
    else:
        c1 = round_width(net.C1_CHANNELS, 2)
        mult = 2
    base = net.C1_CHANNELS * mult
    basis = [(1, base), (2, round_width(base, 2)), (5, round_width(base, 4)),
             (3, round_width(base, 8))]

    stages: List[StageSpec] = []
    out_dim = c1
    counter = 0
    for si, (d, ch) in enumerate(basis):
        in_dim = out_dim
        out_dim = round_width(ch, net.WIDTH_FACTOR)
        inner = int(out_dim * net.BOTTLENECK_WIDTH_FACTOR)
        depth = round_repeats(d, net.DEPTH_FACTOR)
        st = StageSpec(cin=in_dim, inner=inner, cout=out_dim, depth=depth)
        for bi in range(depth):
            counter += 1
            bcin = in_dim if bi == 0 else out_dim
            stride = 2 if bi == 0 else 1
            has_se = (counter + 1) % 2 == 0
            st.blocks.append(BlockSpec(
                stage=si, index=bi, global_index=counter, cin=bcin, inner=inner, cout=out_dim,
                stride=stride, has_se=has_se,
                se_width=round_width(inner, 0.0625) if has_se else 0,
                has_shortcut_conv=(bcin != out_dim or stride != 1)))
        stages.append(st)

    return Arch(
        num_classes=net.NUM_CLASSES, c1=c1, c1_temp_filter=net.C1_TEMP_FILTER, stages=stages,
        conv5_out=stages[-1].inner, fc1_out=2048, bn_eps=float(net.BN.EPS),
        bn_momentum=float(net.BN.MOMENTUM), dropout_rate=float(net.DROPOUT_RATE),
        weight_decay=float(net.WEIGHT_DECAY),
        num_preds=cfg.TEST.NUM_TEMPORAL_VIEWS * cfg.TEST.NUM_SPATIAL_CROPS)


# --------------------------------------------------------------------------------------
# Parameter inventory.  Names are the reference's checkpoint object-graph paths
# (models/X3D-*/model.index) without the '/.ATTRIBUTES/VARIABLE_VALUE' suffix.
# Shapes are the build's native layouts:
#   pointwise kernel  [Cout, Cin]            (TF [1,1,1,Cin,Cout])
#   depthwise 3x3x3   [C, 3, 3, 3]           (TF [3,3,3,1,C])
#   stem conv_s       [Cout, Cin, 3, 3]      (TF [1,3,3,Cin,Cout])
#   stem conv_t       [C, kt]                (TF [kt,1,1,1,C])
#   fc2 kernel        [num_classes, 2048]    (TF Dense [2048, num_classes])
# --------------------------------------------------------------------------------------

@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    kind: str                 # pw | dw | stem_s | stem_t | dense | bias | gamma | beta | mean | var
    trainable: bool
    l2: bool                  # carries the 5e-5 L2 regulariser (model.py:47; se_fc1 has none, :278-283)
    fan_in: int = 0
    fan_out: int = 0


def _bn(prefix, c, gamma="gamma", sep="/"):
    return [ParamSpec(f"{prefix}{sep}gamma", (c,), "gamma", True, False),
            ParamSpec(f"{prefix}{sep}beta", (c,), "beta", True, False),
            ParamSpec(f"{prefix}{sep}moving_mean", (c,), "mean", False, False),
            ParamSpec(f"{prefix}{sep}moving_variance", (c,), "var", False, False)]


def block_prefix(b: BlockSpec) -> str:
    return f"stages/{b.stage}/stage/layer_with_weights-{b.index}"


def param_specs(arch: Arch, in_channels: int = 3) -> List[ParamSpec]:
    """All variables of the model in the reference's creation order per layer."""
    P: List[ParamSpec] = []
    kt = arch.c1_temp_filter
    P.append(ParamSpec("conv1/conv_s/kernel", (arch.c1, in_channels, 3, 3), "stem_s", True, True,
                       fan_in=9 * in_channels, fan_out=9 * arch.c1))
    P.append(ParamSpec("conv1/conv_t/kernel", (arch.c1, kt), "stem_t", True, True,
                       fan_in=kt, fan_out=kt * arch.c1))
    P += _bn("conv1/bn", arch.c1)
    for b in arch.blocks:
        p = block_prefix(b)
        if b.has_shortcut_conv:
            P.append(ParamSpec(f"{p}/residual/kernel", (b.cout, b.cin), "pw", True, True,
                               fan_in=b.cin, fan_out=b.cout))
            P += _bn(f"{p}/bn_r", b.cout)
        q = f"{p}/bottleneck"
        P.append(ParamSpec(f"{q}/a/kernel", (b.inner, b.cin), "pw", True, True,
                           fan_in=b.cin, fan_out=b.inner))
        P += _bn(f"{q}/bn_a", b.inner)
        P.append(ParamSpec(f"{q}/b/kernel", (b.inner, 3, 3, 3), "dw", True, True,
                           fan_in=27, fan_out=27 * b.inner))
        P += _bn(f"{q}/bn_b", b.inner)
        if b.has_se:
            P.append(ParamSpec(f"{q}/se_fc1/kernel", (b.se_width, b.inner), "pw", True, False,
                               fan_in=b.inner, fan_out=b.se_width))
            P.append(ParamSpec(f"{q}/se_fc1/bias", (b.se_width,), "bias", True, False))
            P.append(ParamSpec(f"{q}/se_fc2/kernel", (b.inner, b.se_width), "pw", True, True,
                               fan_in=b.se_width, fan_out=b.inner))
            P.append(ParamSpec(f"{q}/se_fc2/bias", (b.inner,), "bias", True, False))
        P.append(ParamSpec(f"{q}/c/kernel", (b.cout, b.inner), "pw", True, True,
                           fan_in=b.inner, fan_out=b.cout))
        P += _bn(f"{q}/bn_c", b.cout)
    last = arch.stages[-1].cout
    P.append(ParamSpec("conv5/layer_with_weights-0/kernel", (arch.conv5_out, last), "pw", True, True,
                       fan_in=last, fan_out=arch.conv5_out))
    P += _bn("conv5/layer_with_weights-1", arch.conv5_out)
    P.append(ParamSpec("fc1/kernel", (arch.fc1_out, arch.conv5_out), "pw", True, True,
                       fan_in=arch.conv5_out, fan_out=arch.fc1_out))
    P.append(ParamSpec("fc2/kernel", (arch.num_classes, arch.fc1_out), "dense", True, True,
                       fan_in=arch.fc1_out, fan_out=arch.num_classes))
    P.append(ParamSpec("fc2/bias", (arch.num_classes,), "bias", True, False))
    return P


def count_params(arch: Arch, in_channels: int = 3):
    """(total, trainable, non_trainable) -- the footer of the reference's Keras summaries."""
    tot = tr = 0
    for s in param_specs(arch, in_channels):
        n = 1
        for d in s.shape:
            n *= d
        tot += n
        tr += n if s.trainable else 0
    return tot, tr, tot - tr


def summary_rows(arch: Arch, t: int, h: int, w: int, in_channels: int = 3):
    """Rows of reference ``X3D.summary`` (models/X3D-*/X3D_*.txt:5-25): (name, NTHWC shape, params)."""
    specs = param_specs(arch, in_channels)

    def psum(prefix):
        tot = 0
        for s in specs:
            if s.name.startswith(prefix):
                n = 1
                for d in s.shape:
                    n *= d
                tot += n
        return tot

    rows = []
    hh = (h + 2 - 3) // 2 + 1
    ww = (w + 2 - 3) // 2 + 1
    rows.append(("conv_1", (t, hh, ww, arch.c1), psum("conv1/")))
    for i, s in enumerate(arch.stages):
        hh = same_pad(hh, 3, 2)[0]
        ww = same_pad(ww, 3, 2)[0]
        rows.append((f"res_stage_{i + 2}", (t, hh, ww, s.cout), psum(f"stages/{i}/")))
    rows.append(("conv_5", (t, hh, ww, arch.conv5_out), psum("conv5/")))
    rows.append(("pool_5", (1, 1, 1, arch.conv5_out), 0))
    rows.append(("fc_1", (1, 1, 1, arch.fc1_out), psum("fc1/")))
    rows.append(("dropout", (1, 1, 1, arch.fc1_out), 0))
    rows.append(("fc_2", (1, 1, 1, arch.num_classes), psum("fc2/")))
    return rows


# --------------------------------------------------------------------------------------
# Byte / FLOP accounting (SURVEY 8d convention): each conv reads its input once and writes its
# output once; BN / activation / SE scale are folded; the residual tail costs 2 reads + 1 write
# of the block output; weights ignored.  Training step = 3 x forward.
# --------------------------------------------------------------------------------------

def workload(arch: Arch, t: int, h: int, w: int, in_channels: int = 3):
    """Per-clip forward algorithmic element count and FLOPs, split by op family."""
    el = dict(stem_s=0, depthwise=0, pointwise=0, tail=0)
    fl = dict(stem_s=0, depthwise=0, pointwise=0)
    hh = (h + 2 - 3) // 2 + 1
    ww = (w + 2 - 3) // 2 + 1
    p_in = t * h * w
    p = t * hh * ww
    el["stem_s"] += in_channels * p_in + arch.c1 * p
    fl["stem_s"] += 2 * 9 * in_channels * arch.c1 * p
    el["depthwise"] += 2 * arch.c1 * p
    fl["depthwise"] += 2 * arch.c1_temp_filter * arch.c1 * p
    for b in arch.blocks:
        ho = same_pad(hh, 3, b.stride)[0]
        wo = same_pad(ww, 3, b.stride)[0]
        po = t * ho * wo
        el["pointwise"] += (b.cin + b.inner) * p            # a
        fl["pointwise"] += 2 * b.cin * b.inner * p
        el["depthwise"] += b.inner * (p + po)               # b
        fl["depthwise"] += 2 * 27 * b.inner * po
        el["pointwise"] += (b.inner + b.cout) * po          # c
        fl["pointwise"] += 2 * b.inner * b.cout * po
        if b.has_shortcut_conv:
            el["pointwise"] += (b.cin + b.cout) * po
            fl["pointwise"] += 2 * b.cin * b.cout * po
        el["tail"] += 3 * b.cout * po
        hh, ww, p = ho, wo, po
    last = arch.stages[-1].cout
    el["pointwise"] += (last + arch.conv5_out) * p
    fl["pointwise"] += 2 * last * arch.conv5_out * p
    el["pointwise"] += arch.conv5_out + arch.fc1_out
    fl["pointwise"] += 2 * arch.conv5_out * arch.fc1_out
    return dict(elements=el, flops=fl, total_elements=sum(el.values()),
                total_flops=sum(fl.values()))
