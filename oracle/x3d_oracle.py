"""CPU oracle: a PyTorch-CPU (fp32 / fp64) restatement of the reference's X3D graph.

TEST INFRASTRUCTURE ONLY.  Nothing under x3d-tf_amd/ imports this; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may.  It is the checker, never the
product path.

PARITY UNPINNED: the reference ships no tests, golden vectors or known-answer fixtures for the
model path, TensorFlow 2.4.1 is not installable here and the released checkpoints' data shards
are absent.  What *is* pinned: layer structure, shapes and parameter counts against
models/X3D-*/X3D_*.txt, and variable names/shapes against models/X3D-*/model.index
(tests/test_arch.py, tests/test_checkpoint.py).  Numerics follow the rules below, each citing the
reference line it restates; rules that live inside TensorFlow are marked [TF-3p].

Layout: the module boundary is NTHWC like the reference (model.py:113); internally NCTHW.
Parameters are a dict name -> tensor in the native layouts documented in x3d-tf_amd/arch.py.
"""
import math

import torch
import torch.nn.functional as F


def _same_pad(in_size, kernel, stride):
    # [TF-3p] SAME rule: out = ceil(in/s); total = max((out-1)*s + k - in, 0); before = total//2
    out = -(-in_size // stride)
    total = max((out - 1) * stride + kernel - in_size, 0)
    return out, total // 2, total - total // 2


class BNState:
    """Collects what a training-mode forward produces besides activations."""

    def __init__(self):
        self.new_moving = {}      # name -> tensor (moving_mean / moving_variance after the update)
        self.batch_stats = {}     # bn prefix -> (mean, biased var)


def batch_norm(x, p, prefix, training, eps, momentum, state=None, unbiased_moving_var=True):
    """Keras BatchNormalization(axis=-1, epsilon, momentum) on NCTHW x (reference model.py:89-92,
    196-199,254-257,268-271,300-303,368-371).

    training: normalise with the batch mean and *biased* batch variance over N,T,H,W;
      moving <- moving*momentum + batch*(1-momentum)  (Keras momentum convention).  The moving
      variance is fed the unbiased estimate, as TF's fused kernel does for 4-D/5-D inputs
      [TF-3p, unverified for 2.4.1 5-D; differs from biased by M/(M-1), M >= 1e4 here].
    inference: normalise with the moving statistics.
    """
    g, b = p[f"{prefix}/gamma"], p[f"{prefix}/beta"]
    shape = (1, -1, 1, 1, 1)
    if training:
        dims = (0, 2, 3, 4)
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        if state is not None:
            m = x.numel() // x.shape[1]
            var_m = var * (m / max(m - 1, 1)) if unbiased_moving_var else var
            with torch.no_grad():
                state.new_moving[f"{prefix}/moving_mean"] = (
                    p[f"{prefix}/moving_mean"] * momentum + mean.detach() * (1 - momentum))
                state.new_moving[f"{prefix}/moving_variance"] = (
                    p[f"{prefix}/moving_variance"] * momentum + var_m.detach() * (1 - momentum))
                state.batch_stats[prefix] = (mean.detach().clone(), var.detach().clone())
    else:
        mean, var = p[f"{prefix}/moving_mean"], p[f"{prefix}/moving_variance"]
    inv = torch.rsqrt(var + eps)
    return (x - mean.view(shape)) * (inv * g).view(shape) + b.view(shape)


def pointwise(x, w, stride=1):
    """1x1x1 Conv3D, no bias; stride (1,s,s) 'valid' samples pixels 0,s,2s,... (model.py:360-367)."""
    if stride != 1:
        x = x[:, :, :, ::stride, ::stride]
    return torch.einsum("oc,ncthw->nothw", w, x)


def depthwise3x3x3(x, w, stride):
    """Conv3D(k=3x3x3, strides=(1,s,s), padding='same', groups=C) (model.py:259-267)."""
    c = x.shape[1]
    _, hb, ha = _same_pad(x.shape[3], 3, stride)
    _, wb, wa = _same_pad(x.shape[4], 3, stride)
    x = F.pad(x, (wb, wa, hb, ha, 1, 1))
    return F.conv3d(x, w.view(c, 1, 3, 3, 3), stride=(1, stride, stride), groups=c)


class RecordMasks(dict):
    """Pass as `relu_masks` to RECORD the free-running sign pattern (z > 0) of every ReLU site instead of imposing one:
    tests compare it with the device's (tests/util.hip_relu_masks) and bound the number of differing bits."""


def _relu(z, site, masks):
    """ReLU; with `masks` (site -> bool tensor) the given sign pattern is used instead of z > 0.
    ReLU' is discontinuous at 0, so a pre-activation of +-1e-7 that two correct fp32 implementations round
    to opposite signs changes the gradient by O(1/batch elements).  Parity tests therefore hand the oracle
    the masks the device forward used; forward values are unaffected beyond ~1e-6."""
    if isinstance(masks, RecordMasks):
        masks[site] = (z.detach() > 0)
        return F.relu(z)
    if masks is not None and site in masks:
        return z * masks[site].to(z.dtype)
    return F.relu(z)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


class _DwOperands(torch.autograd.Function):
    """depthwise3x3x3 whose forward and backward products each see their operands rounded to `dt_f` / `dt_b` (None: fp32)."""

    @staticmethod
    def forward(ctx, a, w, stride, dt_f, dt_b):
        ctx.save_for_backward(a, w)
        ctx.stride, ctx.dt_b = stride, dt_b
        r = (lambda v: v.to(dt_f).to(v.dtype)) if dt_f is not None else (lambda v: v)
        return depthwise3x3x3(r(a), r(w), stride)

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        dt = ctx.dt_b
        r = (lambda v: v.to(dt).to(v.dtype)) if dt is not None else (lambda v: v)
        with torch.enable_grad():
            a_ = r(a.detach()).requires_grad_(True)
            w_ = r(w.detach()).requires_grad_(True)
            y = depthwise3x3x3(a_, w_, ctx.stride)
            da, dw = torch.autograd.grad(y, [a_, w_], r(g))
        return da, dw, None, None, None


class Storage:
    """Emulates a reduced-precision HBM storage type on top of the fp32 graph.

    The device path computes in fp32 and only *stores* activations and activation gradients in bf16.
    ``act(x)`` rounds a tensor where the device writes it to HBM (raw conv outputs, block outputs);
    ``grad(x)`` rounds the gradient flowing into x where the device writes that gradient (dy/g, dv, ga,
    the strided-shortcut temporary, g5, ds).  Batch statistics are then taken from the rounded tensors,
    exactly as the kernels' epilogues do.  With dtype None both are the identity (pure fp32 oracle).
    Deep random-init BN networks amplify perturbations by ~600x end to end (measured: fp32 rounding noise
    1e-7 -> 7e-5 gradient error), so bf16 parity can only be stated against this storage-faithful variant.
    """

    def __init__(self, dtype=None, dw_operands=None):
        self.dtype = None if dtype in (None, torch.float32) else dtype
        # site (block prefix) -> (forward on the matrix cores?, backward on the matrix cores?): the depthwise kernels of
        # dw_mx.hip multiply operands ROUNDED to the storage type (relu(bn_a(a)), dB, the 27 weights), the vector kernels
        # multiply in fp32 -- and the two directions of one layer may run different kernels.  Tests fill this from the
        # library's own dispatch (x3d_dw3d_kernel_name); empty = fp32 operands everywhere.
        self.dw_operands = dw_operands or {}

    def act(self, x):
        return x if self.dtype is None else _RoundFwd.apply(x, self.dtype)

    def grad(self, x):
        return x if self.dtype is None else _RoundBwd.apply(x, self.dtype)

    def depthwise(self, a_act, w, stride, site):
        """the 3x3x3 channelwise conv with the operand rounding of the kernels that run it on the device"""
        fwd_mx, bwd_mx = self.dw_operands.get(site, (False, False))
        if self.dtype is None or not (fwd_mx or bwd_mx):
            return depthwise3x3x3(a_act, w, stride)
        return _DwOperands.apply(a_act, w, stride, self.dtype if fwd_mx else None, self.dtype if bwd_mx else None)


_FP32 = Storage(None)


def stem(x, p, arch, training, state, masks=None, st=_FP32):
    """X3D_Stem.call (model.py:202-210): pad(0,1,1) -> conv_s 1x3x3 s(1,2,2) valid -> pad(kt//2,0,0)
    -> conv_t ktx1x1 depthwise -> BN -> ReLU.  No BN/activation between the two convs."""
    ws = p["conv1/conv_s/kernel"]                       # [Cout, Cin, 3, 3]
    wt = p["conv1/conv_t/kernel"]                       # [C, kt]
    kt = wt.shape[1]
    y = st.grad(st.act(F.conv3d(F.pad(x, (1, 1, 1, 1, 0, 0)), ws.unsqueeze(2), stride=(1, 2, 2))))
    y = st.act(F.conv3d(F.pad(y, (0, 0, 0, 0, kt // 2, kt // 2)), wt.view(-1, 1, kt, 1, 1),
                        groups=wt.shape[0]))
    y = batch_norm(y, p, "conv1/bn", training, arch.bn_eps, arch.bn_momentum, state)
    return st.grad(st.act(_relu(y, "conv1", masks)))


def block_prefix(b):
    return f"stages/{b.stage}/stage/layer_with_weights-{b.index}"


def res_block(x, p, b, arch, training, state, taps=None, masks=None, st=_FP32):
    """ResBlock.call (model.py:384-394) around Bottleneck.call (model.py:305-320)."""
    pre = block_prefix(b)
    q = f"{pre}/bottleneck"
    eps, mom = arch.bn_eps, arch.bn_momentum
    a = st.act(pointwise(x, p[f"{q}/a/kernel"]))
    # st.grad: the device stores ga = d loss / d bn_a(a) (ReLU mask already applied)
    a_act = _relu(st.grad(batch_norm(a, p, f"{q}/bn_a", training, eps, mom, state)), f"{pre}/a", masks)
    bb = st.act(st.depthwise(a_act, p[f"{q}/b/kernel"], b.stride, pre))
    u = batch_norm(bb, p, f"{q}/bn_b", training, eps, mom, state)
    if b.has_se:
        # SE (model.py:274-290,311-315): global mean -> fc1(+bias, ReLU) -> fc2(+bias, sigmoid) -> scale,
        # applied after BN_b and before swish.
        pooled = u.mean((2, 3, 4))                                          # [N, C]
        s1 = _relu(pooled @ p[f"{q}/se_fc1/kernel"].t() + p[f"{q}/se_fc1/bias"], f"{pre}/se", masks)
        gate = torch.sigmoid(s1 @ p[f"{q}/se_fc2/kernel"].t() + p[f"{q}/se_fc2/bias"])
        u = u * gate[:, :, None, None, None]
    u = st.grad(u)                                                          # device stores dv = d loss / d (gate*bn_b(b))
    s = u * torch.sigmoid(u)                                                # tf.nn.swish
    c = st.act(pointwise(s, p[f"{q}/c/kernel"]))
    c = batch_norm(c, p, f"{q}/bn_c", training, eps, mom, state)
    if b.has_shortcut_conv:
        xs = x[:, :, :, ::b.stride, ::b.stride] if b.stride != 1 else x
        r = st.act(pointwise(st.grad(xs), p[f"{pre}/residual/kernel"]))    # st.grad: the strided dgrad temporary
        r = batch_norm(r, p, f"{pre}/bn_r", training, eps, mom, state)
    else:
        r = x
    # block output stored once; its gradient (dy, then g = dy*[y>0] in place) stored once
    y = st.grad(st.act(_relu(r + c, f"{pre}/out", masks)))
    if taps is not None:
        taps[f"{pre}/a_raw"] = a
        taps[f"{pre}/b_raw"] = bb
        taps[f"{pre}/out"] = y
    return y


def forward(p, x_nthwc, arch, training=False, dropout_mask=None, state=None, taps=None,
            return_logits=False, relu_masks=None, storage=None):
    """X3D.call (model.py:113-127).  x_nthwc: [N,T,H,W,3].  Returns fp32 probabilities
    [N, classes] in training mode and [N / num_preds, classes] (view-averaged) otherwise.

    storage: None/float32 = the pure fp32 graph; torch.bfloat16 = additionally round every tensor (and
    gradient) the device path stores in HBM (see Storage).

    dropout_mask: optional [N, 2048] tensor of 0/1 keep flags (training only); kept units are
    scaled by 1/(1-rate) [TF-3p Dropout].  None with training=True and rate>0 draws one.
    """
    st = storage if isinstance(storage, Storage) else Storage(storage)
    x = st.act(x_nthwc.permute(0, 4, 1, 2, 3))
    out = stem(x, p, arch, training, state, relu_masks, st)
    if taps is not None:
        taps["conv1/out"] = out
    for b in arch.blocks:
        out = res_block(out, p, b, arch, training, state, taps, relu_masks, st)
    out = st.act(pointwise(out, p["conv5/layer_with_weights-0/kernel"]))
    out = _relu(st.grad(batch_norm(out, p, "conv5/layer_with_weights-1", training, arch.bn_eps,
                                   arch.bn_momentum, state)), "conv5", relu_masks)
    pooled = out.mean((2, 3, 4))                                            # pool5 (model.py:118)
    h = _relu(pooled @ p["fc1/kernel"].t(), "fc1", relu_masks)                                # fc1: no bias (model.py:95-102)
    if training and arch.dropout_rate > 0:
        if dropout_mask is None:
            dropout_mask = (torch.rand_like(h) >= arch.dropout_rate).to(h.dtype)
        h = h * dropout_mask / (1.0 - arch.dropout_rate)
    logits = h @ p["fc2/kernel"].t() + p["fc2/bias"]
    probs = torch.softmax(logits.float(), -1)                               # fp32 softmax (model.py:111)
    if taps is not None:
        taps["logits"] = logits
    if not training:
        # average the views of each video (model.py:123-126): rows are grouped consecutively
        if probs.shape[0] % arch.num_preds:
            raise ValueError(
                f"inference batch {probs.shape[0]} is not a multiple of views*crops={arch.num_preds}")
        probs = probs.view(-1, arch.num_preds, probs.shape[-1]).mean(1)
        if return_logits:
            return probs, logits
        return probs
    if return_logits:
        return probs, logits
    return probs.reshape(-1, arch.num_classes)


def l2_names(p):
    """Kernels carrying the L2 regulariser (model.py:47; every conv/dense kernel except se_fc1,
    model.py:278-283; never BN parameters or biases)."""
    return [k for k in p if k.endswith("/kernel") and "/se_fc1/" not in k]


def loss_fn(probs, labels, p, arch):
    """tf.keras.losses.SparseCategoricalCrossentropy() on *probabilities* (train.py:104) + L2.

    [TF-3p] Keras' backend clips p to [1e-7, 1-1e-7], takes the log and feeds it as logits to the
    sparse softmax cross-entropy, i.e. loss_i = -log q_y + log sum_j q_j with q = clip(p); mean over
    the batch.  The L2 term is weight_decay * sum(w^2) per regularised kernel (no 1/2 factor).
    """
    q = probs.clamp(1e-7, 1.0 - 1e-7)
    ce = (-torch.log(q.gather(1, labels.view(-1, 1).long()).squeeze(1)) + torch.log(q.sum(1))).mean()
    reg = sum((p[k].double() ** 2).sum() for k in l2_names(p)).to(probs.dtype) * arch.weight_decay
    return ce + reg, ce, reg


def sgd_nesterov_(p, grads, velocity, lr, momentum):
    """tf.optimizers.SGD(momentum, nesterov=True) (train.py:89-92) [TF-3p update rule]:
    v <- m*v - lr*g ;  w <- w + m*v - lr*g."""
    with torch.no_grad():
        for k, g in grads.items():
            v = velocity[k]
            v.mul_(momentum).sub_(lr * g)
            p[k].add_(momentum * v - lr * g)


def lr_schedule(epoch, cfg):
    """train.py:114-125: linear warm-up for epoch <= WARMUP_EPOCHS, then half-cosine; per epoch."""
    tr = cfg.TRAIN
    if epoch > tr.WARMUP_EPOCHS:
        return tr.BASE_LR * (0.5 * (math.cos(math.pi * (epoch / tr.EPOCHS)) + 1))
    return tr.WARMUP_LR + epoch * (tr.BASE_LR - tr.WARMUP_LR) / tr.WARMUP_EPOCHS


def trainable_names(p):
    return [k for k in p if not (k.endswith("/moving_mean") or k.endswith("/moving_variance"))]


def train_step(p, x_nthwc, labels, arch, velocity=None, lr=None, momentum=0.9, dropout_mask=None,
               apply_update=True, relu_masks=None, taps=None, storage=None):
    """One fwd+bwd(+SGD) step.  Returns dict(loss, ce, reg, probs, grads, state)."""
    names = trainable_names(p)
    leaf = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    state = BNState()
    probs = forward(leaf, x_nthwc, arch, training=True, dropout_mask=dropout_mask, state=state,
                    relu_masks=relu_masks, taps=taps, storage=storage)
    loss, ce, reg = loss_fn(probs, labels, leaf, arch)
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = dict(zip(names, gl))
    if apply_update and lr is not None:
        if velocity is None:
            velocity = {k: torch.zeros_like(p[k]) for k in names}
        sgd_nesterov_(p, grads, velocity, lr, momentum)
        for k, v in state.new_moving.items():
            p[k].copy_(v)
    return dict(loss=loss.detach(), ce=ce.detach(), reg=reg.detach(), probs=probs.detach(),
                grads=grads, state=state, velocity=velocity)
