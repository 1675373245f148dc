"""Whole-model parity of the HIP path against the CPU oracle on the same seeded inputs and weights.

fp32: `X3D.call` output (softmax probabilities) within 1e-4 abs of the oracle -- the
tolerance BASELINE.json config 2 states -- plus pre-softmax logits, per-block activations, the loss,
every one of the 308 parameter gradients, the BN moving statistics and the Nesterov update.
bf16: loose tolerances on the same quantities (bf16 storage of the activations, fp32 arithmetic).

ReLU' is discontinuous: a pre-activation of 1e-7 that two correct fp32 evaluations round to opposite
signs changes gradients by O(1/elements-per-channel) (measured: 5 such elements of 1.5 M at XS 4x64x64
move late-stage gradients by 1e-2).  The gradient checks therefore hand the oracle the ReLU sign
patterns of the device forward (tests/util.hip_relu_masks); forward parity is checked free-running.
"""
import json
import os

import pytest
import torch

from tests import shapes as S
from tests.util import hip_relu_masks, rel_l2, relu_mask_mismatch, report
from x3d_tf_amd.arch import block_prefix

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(name, overrides=None):
    import x3d_tf_amd as x
    from x3d_tf_amd.params import init_params, randomize_bn_
    cfg = x.get_config(name, overrides)
    arch = x.build_arch(cfg)
    params = randomize_bn_(init_params(arch, seed=3), seed=4)
    return cfg, arch, params


def _model(cfg, params, dtype, gpu, options=None):
    from x3d_tf_amd.model import X3D
    m = X3D(cfg, dtype=dtype, device=gpu, seed=0, options=options)
    m.load_state_dict(params)
    return m


def _scaled(name, got, ref, rel):
    """max abs error relative to the tensor's own magnitude"""
    scale = ref.detach().abs().max().item() + 1e-30
    return report(name, got, ref, 0, rel * scale)


@pytest.mark.parametrize("name,views,crops,t,s,dtype", S.MODEL_INFER)
def test_forward_inference(gpu, name, views, crops, t, s, dtype):
    """Inference mode with view averaging.  BASELINE config 1: X3D-XS, one video = 10 views of 4x160x160 (fp32, 1e-4);
    BASELINE config 5: X3D-XL, 10 temporal views x 3 spatial crops = 30 clips per video, fp32 and the 16-bit storage
    types (fp16 is the reference's mixed_float16).  No batch statistics in this mode, so the 16-bit runs are compared
    end to end: logits within 2 % (bf16) / 0.5 % (fp16) of the largest logit, probabilities within 5e-4 / 1.5e-4."""
    from oracle import x3d_oracle as O
    cfg, arch, params = _setup(name, ["TEST.NUM_TEMPORAL_VIEWS", views, "TEST.NUM_SPATIAL_CROPS", crops])
    nv = views * crops
    assert arch.num_preds == nv
    torch.manual_seed(0)
    x = torch.randn(nv, t, s, s, 3)
    if dtype != torch.float32:
        x = x.to(dtype).float()
    ref, ref_logits = O.forward(params, x, arch, training=False, return_logits=True)
    m = _model(cfg, params, dtype, gpu)
    out = m(x.to(gpu), training=False)
    torch.cuda.synchronize()
    assert out.dtype == torch.float32 and tuple(out.shape) == (1, arch.num_classes)
    pl = m._plans[(nv, t, s, s, False)]
    if dtype == torch.float32:
        _scaled("logits", pl.logits, ref_logits, 1e-4)
        report("probs", out, ref, 0, 1e-4)
    else:
        rel, pabs = (2e-2, 5e-4) if dtype == torch.bfloat16 else (5e-3, 1.5e-4)
        _scaled("logits", pl.logits, ref_logits, rel)
        report("probs", out, ref, 0, pabs)
    assert abs(out.sum().item() - 1.0) < 1e-5


def test_forward_matches_committed_golden_vector(gpu):
    """tests/golden/oracle_xs_forward.json: the oracle's X3D-XS 10-view inference, generated in the build
    container by tests/golden/make_golden.py (seeds recorded in the file)."""
    gold = json.load(open(os.path.join(GOLDEN, "oracle_xs_forward.json")))
    cfg, arch, params = _setup(gold["config"])
    torch.manual_seed(gold["input_seed"])
    x = torch.randn(*gold["input_shape"])
    m = _model(cfg, params, torch.float32, gpu)
    out = m(x.to(gpu), training=False)
    pl = m._plans[tuple(gold["input_shape"][:4]) + (False,)]
    ref_logits = torch.tensor(gold["logits_view0"])
    _scaled("logits_view0", pl.logits[0], ref_logits, 1e-4)
    report("probs", out[0], torch.tensor(gold["probs"]), 0, 1e-4)


def test_input_batch_is_read_where_it_lies(gpu):
    """16-bit storage: the stem reads the caller's channels-last batch in place (x3d_hip.h K1, X3D_LAYOUT_NTHWC).  The same
    clips handed over as a fresh tensor, as a 2-byte-offset view (copied once: the kernels need 16-byte alignment), in fp32
    (converted once) and through the planar fallback (options stem_nthwc = False) give the same probabilities bit for bit; the plan
    holds no planar copy of the batch."""
    cfg, arch, params = _setup("XS", ["TEST.NUM_TEMPORAL_VIEWS", 2, "TEST.NUM_SPATIAL_CROPS", 1])
    torch.manual_seed(5)
    x = torch.randn(2, 4, 64, 64, 3).to(torch.bfloat16)
    m = _model(cfg, params, torch.bfloat16, gpu)
    ref = m(x.to(gpu), training=False).clone()
    pl = m._plans[(2, 4, 64, 64, False)]
    assert pl.x_cl and pl.x is None
    base = torch.zeros(x.numel() + 8, dtype=torch.bfloat16, device=gpu)
    off = base[1:1 + x.numel()].view(x.shape)
    off.copy_(x.to(gpu))
    assert off.data_ptr() % 16 != 0
    assert torch.equal(m(off, training=False), ref)
    assert torch.equal(m(x.float().to(gpu), training=False), ref)
    m2 = _model(cfg, params, torch.bfloat16, gpu, options={"stem_nthwc": False})
    out2 = m2(x.to(gpu), training=False)
    assert not m2._plans[(2, 4, 64, 64, False)].x_cl
    assert torch.equal(out2, ref)


def test_inference_batch_must_be_multiple_of_views(gpu):
    cfg, arch, params = _setup("XS")
    m = _model(cfg, params, torch.float32, gpu)
    with pytest.raises(ValueError):
        m(torch.randn(3, 4, 32, 32, 3, device=gpu), training=False)


@pytest.mark.parametrize("name,n,t,s", S.MODEL_TRAIN_FP32)
def test_train_step_fp32(gpu, name, n, t, s):
    """fwd + bwd in training mode (batch statistics, fixed dropout mask) against oracle autograd."""
    from oracle import x3d_oracle as O
    cfg, arch, params = _setup(name)
    torch.manual_seed(1)
    x = torch.randn(n, t, s, s, 3)
    labels = torch.randint(0, arch.num_classes, (n,))
    mask = (torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float()

    m = _model(cfg, params, torch.float32, gpu)
    m.set_dropout_mask(mask)
    pl = m.forward_backward(x.to(gpu), labels.to(gpu))
    torch.cuda.synchronize()

    # free-running oracle forward: activations and probabilities
    taps = {}
    st = O.BNState()
    free_masks = O.RecordMasks()
    probs_free = O.forward({k: v.clone() for k, v in params.items()}, x, arch, training=True, dropout_mask=mask,
                           state=st, taps=taps, relu_masks=free_masks)
    _scaled("conv1/out", pl.y0, taps["conv1/out"], 2e-5)
    for B in pl.blocks:
        pre = O.block_prefix(B.spec)
        _scaled(pre + "/a_raw", B.a_raw, taps[pre + "/a_raw"], 2e-4)
        _scaled(pre + "/b_raw", B.b_raw, taps[pre + "/b_raw"], 2e-4)
        _scaled(pre + "/out", B.y, taps[pre + "/out"], 2e-4)
    _scaled("logits", pl.logits, taps["logits"], 1e-4)
    report("probs", pl.probs, probs_free, 0, 1e-4)
    for k, v in st.new_moving.items():
        report(k, m.params[k], v, 1e-4, 1e-5)

    # gradients and update with the device's ReLU sign patterns -- which may differ from the free-running oracle's only
    # where a pre-activation is within fp32 rounding of zero: at most 1e-5 of the sites (a systematic sign error near
    # zero would show here instead of being absorbed by the hand-over)
    dev_masks = hip_relu_masks(pl)
    frac, bad, tot = relu_mask_mismatch(dev_masks, free_masks)
    assert set(free_masks) == set(dev_masks)
    assert frac <= 1e-5, f"{bad} of {tot} ReLU signs differ between the device and the free-running oracle"
    ref_p = {k: v.clone() for k, v in params.items()}
    r = O.train_step(ref_p, x, labels, arch, lr=0.05, momentum=0.9, dropout_mask=mask, apply_update=True,
                     relu_masks=dev_masks)
    loss = pl.loss_rows.mean() + m.regularization_loss().float()
    report("loss", loss.view(1), r["loss"].view(1), 1e-5, 1e-5)
    for k, g_ref in r["grads"].items():
        g = m.grads[k].cpu().double()
        if m.specs[k].l2:
            g = g + 2 * arch.weight_decay * params[k].double()
        e = rel_l2(g, g_ref)
        assert e < 1e-3, f"grad {k}: relative L2 error {e:.3e}"
        _scaled("grad " + k, g, g_ref, 2e-3)
    m.apply_sgd(0.05, 0.9)
    torch.cuda.synchronize()
    for k, g_ref in r["grads"].items():
        # dw = -lr * (1 + momentum) * g from a zero velocity: the gradient tolerance above (2e-3 of max |g|) carries over
        tol = 1e-4 * ref_p[k].abs().max().item() + 0.05 * 1.9 * 2e-3 * g_ref.abs().max().item()
        report("updated " + k, m.params[k], ref_p[k], 0, tol)


def test_train_matches_committed_golden_vector(gpu):
    """tests/golden/oracle_train_tiny.json: loss and gradient norms of one oracle training step (free-running
    ReLUs, so the tolerance is the ReLU-flip bound, 5 %)."""
    gold = json.load(open(os.path.join(GOLDEN, "oracle_train_tiny.json")))
    cfg, arch, params = _setup(gold["config"])
    torch.manual_seed(gold["input_seed"])
    n = gold["input_shape"][0]
    x = torch.randn(*gold["input_shape"])
    labels = torch.randint(0, 400, (n,))
    assert labels.tolist() == gold["labels"]
    mask = (torch.rand(n, 2048) >= 0.5).float()
    m = _model(cfg, params, torch.float32, gpu)
    m.set_dropout_mask(mask)
    pl = m.forward_backward(x.to(gpu), labels.to(gpu))
    loss = (pl.loss_rows.mean() + m.regularization_loss().float()).item()
    assert abs(loss - gold["loss"]) < 1e-4 * abs(gold["loss"])
    for k, nrm in gold["grad_l2"].items():
        g = m.grads[k].cpu().double()
        if m.specs[k].l2:
            g = g + 2 * arch.weight_decay * params[k].double()
        assert abs(g.norm().item() - nrm) < 5e-2 * nrm, f"{k}: |g| {g.norm().item()} vs {nrm}"


@pytest.mark.parametrize("dtype", S.HALF_DTYPES)
@pytest.mark.parametrize("name,n,t,s", S.MODEL_TRAIN_HALF)
def test_train_step_half_block_by_block(gpu, name, n, t, s, dtype):
    """16-bit activation storage (fp32 accumulation; the operands of the matrix-core products -- pointwise convs, and the
    depthwise planes dw_mx.hip covers -- rounded to the storage type), checked with TEACHER FORCING: every residual block of the
    device run is replayed on the oracle from the device's own stored block input (forward) and the device's
    own stored upstream gradient (backward), with the oracle rounding to bf16 at the tensors the device
    stores (oracle Storage).  End-to-end comparison is meaningless in bf16: a random-init BN network amplifies
    perturbations ~600x (measured in fp32), and bf16 rounding re-injects any 1e-7 difference as a 4e-3 one, so
    two exact implementations decorrelate to O(30 %) on gradients.  Per block the error is one bf16 ulp class:
    stated tolerance 2 % of each tensor's max for activations/gradients; weight gradients (whose GEMM operands
    are rounded to bf16 for the matrix cores) 6.5 % relative L2 worst case (8 % for the two SE biases), 1.5 % median."""
    from oracle import x3d_oracle as O
    cfg, arch, params = _setup(name)
    torch.manual_seed(int(os.environ.get("X3D_TEST_SEED", "2")))      # (the variable: spread of the per-tensor errors over inputs)
    x = torch.randn(n, t, s, s, 3).to(dtype).float()
    labels = torch.randint(0, arch.num_classes, (n,))
    mask = (torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float()
    # (X3D_TEST_RC=0: the same test over the stored-output backward -- the seed sweeps under profiles/ run both)
    m = _model(cfg, params, dtype, gpu, options={"pw_bwd_rc": False} if os.environ.get("X3D_TEST_RC") == "0" else None)
    m.set_dropout_mask(mask)
    # fp16 gradients get the reference's loss scaling (LossScaleOptimizer, train.py:99-100): a power of two, so the oracle
    # replay -- fed the device's own (scaled) upstream gradient per block -- rounds exactly as the device does
    ls = 1024.0 if dtype == torch.float16 else 1.0
    pl = m.forward_backward(x.to(gpu), labels.to(gpu), loss_scale=ls)        # builds the plan, full step
    torch.cuda.synchronize()
    assert torch.isfinite(pl.loss_rows).all() and torch.isfinite(m.flat_grads).all()
    # which depthwise launches run on the matrix cores (operands rounded to the storage type): the library's own dispatch
    from x3d_tf_amd import hip as _hip
    dw_ops = {O.block_prefix(B.spec): ("_mx" in _hip.dw3d_kernel_name(B.sb), "_mx" in _hip.dw3d_kernel_name(B.db))
              for B in pl.blocks}
    st = O.Storage(dtype, dw_operands=dw_ops)
    # replay the backward block by block, capturing each block's upstream gradient before it is overwritten
    m.flat_grads.zero_()
    pl.zero_buf.zero_()
    pl.run(pl.fwd, 0, pl.grad_scale_slot)
    from x3d_tf_amd import hip
    hip.call("x3d_softmax_xent", pl.logits.data_ptr(), pl.labels.data_ptr(), pl.probs.data_ptr(),
             pl.loss_rows.data_ptr(), pl.dlogits.data_ptr(), ls / n, n, arch.num_classes)
    torch.cuda.synchronize()
    # the ReLU sign patterns of THIS forward pass (the one whose stored tensors the blocks are replayed from): two passes of
    # the same plan agree to the last fp32 bit of their batch statistics only when their reductions run in the same order,
    # and one ulp there re-rounds 16-bit tensors downstream -- measured (tools/debug_replay.py): first pass and replay of
    # X3D-M 2x4x128 differ in 10-20 elements of a stage-3 tensor and in hundreds of last-stage signs when another model's
    # buffers are alive in the process.  The backward launches do not write forward tensors.
    masks = hip_relu_masks(pl)
    pos = 0
    worst = dict(y=0.0, dx=0.0, dw=0.0)
    errs = {}
    pending = []
    folded = 0
    for bi in range(len(pl.blocks) - 1, -1, -1):
        B = pl.blocks[bi]
        pl.run(pl.bwd, pos, B.bwd_start)
        torch.cuda.synchronize()
        dy = B.dy_view.float().cpu().clone()       # carries the loss scale, like everything downstream of it
        pl.run(pl.bwd, B.bwd_start, B.bwd_stop)
        torch.cuda.synchronize()
        pos = B.bwd_stop
        dx_dev = B.dx_view.float().cpu()
        # oracle on the device's block input
        pre = O.block_prefix(B.spec)
        names = [k for k in params if k.startswith(pre + "/") and not k.endswith(("moving_mean", "moving_variance"))]
        leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in params.items()}
        xin = B.x.float().cpu().clone().requires_grad_(True)
        y_ref = O.res_block(xin, leaf, B.spec, arch, True, O.BNState(), None, masks, st)
        atol_ = 2e-2 if dtype == torch.bfloat16 else 3e-3
        worst["y"] = max(worst["y"], _scaled(pre + "/out", B.y, y_ref, atol_))
        grads = torch.autograd.grad(y_ref, [xin] + [leaf[k] for k in names], grad_outputs=dy)
        dx_ref = grads[0]
        if (bi > 0 and pl.blocks[bi - 1].tail_folded) or (bi == 0 and getattr(pl, "stem_bwd_folded", False)):
            # this block's `a` backward already applied the Add + ReLU backward of the block below (whose y is B.x): what it
            # stored is the masked gradient.  (dy above is then masked as well: the oracle's ReLU backward re-applies the same
            # mask, which changes nothing.)  Block 0: the same fold with the stem's ReLU (B.x = y0 = relu(bn(t_raw))).
            dx_ref = dx_ref * (B.x.float().cpu() > 0)
            folded += 1
        _scaled(pre + "/dx", dx_dev, dx_ref, atol_)       # every block: identity shortcuts add dy, conv shortcuts their dgrad
        # relative L2 per tensor, the denominator floored at 5 % of the block's typical gradient magnitude (rms over its
        # tensors): the SE bias / kernel gradients of some blocks are sums that cancel to ~1e-3 of that -- any two correct
        # 16-bit evaluations differ by more than such a remainder (measured: se_fc1/bias of stage 0 block 0 between 1e-2 and
        # 0.8 of ITSELF from run to run of the same code, depending on where the 16-bit roundings of dy fall)
        rms = [g_ref.double().norm().item() / max(g_ref.numel(), 1) ** 0.5 for g_ref in grads[1:]]
        floor_rms = 0.05 * sorted(rms)[len(rms) // 2]
        pending.append((names, [g_.double() for g_ in grads[1:]], floor_rms))
    # the weight gradients are read once the WHOLE list has run: a launch of a later block may finish an earlier block's
    # gradient (the dW of the recomputed-output `a` backward rides on the next BatchNorm-backward finalize launch)
    pl.run(pl.bwd, pos, len(pl.bwd))
    torch.cuda.synchronize()
    for names, refs, floor_rms in pending:
        for k, g_ref in zip(names, refs):
            num = (m.grads[k].detach().double().cpu() - g_ref).norm().item()
            den = max(g_ref.norm().item(), floor_rms * g_ref.numel() ** 0.5)
            e = num / (den + 1e-30)
            worst["dw"] = max(worst["dw"], e)
            errs[k] = e
    print("teacher-forced worst:", name, n, t, s, dtype, "mx (fwd, bwd) blocks:", sum(v[0] for v in dw_ops.values()), sum(v[1] for v in dw_ops.values()), worst, sorted(errs.items(), key=lambda kv: -kv[1])[:4], "tail folded in", folded, "blocks")
    # Limits (round 4, after the oracle learnt which depthwise products see rounded operands -- Storage.dw_operands): measured
    # worst over the six cases 4.9e-2 bf16 / 6.4e-3 fp16 (the bn_a gammas: cancelling sums of 16-bit products); the two SE
    # biases are sums that cancel to ~1e-3 of their terms (docstring above) and sit at 5.3e-2 .. 6.1e-2 / 6.2e-3
    # Seed sweep of the M 1x4x224 bf16 case (X3D_TEST_SEED = 2 .. 7, end of round 4): the worst tensor is always a bn_a gamma and
    # lands between 4.6e-2 and 6.0e-2 depending on the input alone (5.4 / 4.7 / 5.7 / 4.9 / 5.2 / 5.4e-2 before the 48 -> 216
    # layer took the recomputed-output backward, 5.7 / 4.7 / 6.0 / 4.9 / 5.2 / 5.4e-2 after: only the stage-2 tensor downstream
    # of it moves, by 3e-3) -- the bf16 limit below is that spread plus 10 %, not a margin for a kernel error of that size.
    # Round 6 (profiles/r06_seed_sweep.log: seeds 2 .. 7, recomputed-output backward on and off, at HEAD -- the fp16 depthwise
    # backward on the matrix cores too, 14 blocks, which the round-5 log predated): worst general tensor 6.0e-2 bf16 (a bn_a
    # gamma, seed 4) / 7.2e-3 fp16 (seed 6), worst SE bias 6.3e-2 / below the general worst.  Limits = that spread plus 25 %:
    # the old bf16 limit (6.5e-2) sat 0.3 % above one seed's value.
    lim, lim_se, med = (7.5e-2, 8e-2, 1.5e-2) if dtype == torch.bfloat16 else (9e-3, 1.2e-2, 2.5e-3)
    bad = {k: e for k, e in errs.items() if e > (lim_se if k.endswith(("/se_fc1/bias", "/se_fc2/bias")) else lim)}
    assert not bad, f"relative L2 error beyond {lim} ({lim_se} for the SE biases) (teacher-forced, {dtype}): {bad}"
    assert sorted(errs.values())[len(errs) // 2] < med        # median


# Limits of the differential test below, per storage type: (tensors no recomputed-output launch can reach; every other tensor;
# BatchNorm gammas; the SE branch's first layer and biases).  The first is fp32 atomic summation order only (measured: 0 .. 3e-7).  The others are
# what the fold's extra operand rounding allows: the panel [W^T A | W^T B W] is rounded to the storage type once per step
# (2^-9 relative per entry in bf16, 2^-12 in fp16) and dx = [W1 | M][g ; x] + c0 is a CANCELLING sum (the BatchNorm backward
# removes the mean and the yhat-correlated part of g), so the rounding shows at ~5x its size: test_pw_bwd_rc bounds it at
# 2e-2 / 3e-3 of the tensor maximum per layer, and every gradient downstream inherits it LINEARLY (the forward state is
# shared, so the backward pass is a linear map of the upstream gradient: nothing amplifies).  Measured on one MI355X, seed 2
# (profiles/r05_rc_differential.txt): bf16 general tensors <= 2.1e-2, bn_a gammas (sums that cancel further) <= 4.6e-2, the
# stage-2 block-0 se_fc1 bias (cancels to ~1e-3 of its terms, see the teacher-forced test) 8.6e-2; fp16 3.9e-3 / 4.2e-3 / 3e-3.
# A wrong coefficient table, rc_sums slot or a stale panel is an O(1) error on the tensor it touches.
# With perturbed weights (the second round) the stage-2 SE gradients cancel further: X3D-L block 0 se_fc1 bias 1.7e-1, kernel
# 3.6e-2 -- d(loss)/d(pooled) of an SE branch is a sum over 54 channels whose terms cancel to ~1e-3; the limit of that class is
# set from it and guards against nothing but an O(1) error there.
# Round 6: the limits are the spread of a SEED SWEEP (tools/rc_diff_sweep.py, seeds 2 .. 7, both shapes, both rounds:
# profiles/r06_rc_diff_sweep.log) plus 25 %, no longer one seed's value plus headroom -- the worst tensor of a class is a
# heavy-tailed sample of the operand rounding (another forward state, e.g. another summation order of the stem's batch
# statistics, draws another): bf16 general 3.9e-2 / gammas 6.9e-2 / SE class 1.7e-1, fp16 4.3e-3 / 9.8e-3 / 7.1e-3.  Fifth entry:
# d(loss)/d(pooled) of the SE blocks (1.8e-1 / 5.7e-2: see one_round).
RC_DIFF_LIMITS = {torch.bfloat16: (2e-5, 5e-2, 8.5e-2, 2.5e-1, 2.5e-1), torch.float16: (2e-5, 5.5e-3, 1.25e-2, 3e-2, 8e-2)}


@pytest.mark.parametrize("dtype", S.HALF_DTYPES)
@pytest.mark.parametrize("name,n,t,s", [("M", 1, 4, 224), ("L", 1, 2, 312)])
def test_recomputed_output_backward_against_the_stored_output_backward(gpu, name, n, t, s, dtype):
    """The WIRING of the recomputed-output `a` / shortcut backward (pw_bwd_rc.hip) inside a plan, pinned differentially:
    the same plan, the same forward state, the backward list recorded twice -- plan option pw_bwd_rc on (the product) and
    off (the round-3 kernels that read a_raw / r_raw) -- on the headline's planes (X3D-M 224^2) and config 4's (X3D-L 312^2).
    What the kernel tests cannot see and the teacher-forced test sees only at 6.5e-2: which coefficient table and which
    rc_sums buffer a launch is handed, x3d_bn_bwd_finalize_rc finishing an EARLIER layer's dW on a LATER launch, the panel
    rebuilt per step (the second round below runs after the weights changed: a panel left over from the first step would be
    an O(1) error).  Gradients no recomputed-output launch can reach agree to fp32 summation order (in practice bit for bit);
    all others to the operand-rounding bound stated at RC_DIFF_LIMITS."""
    from tests.util import record_alternate_backward
    from x3d_tf_amd import hip
    cfg, arch, params = _setup(name)
    torch.manual_seed(int(os.environ.get("X3D_TEST_SEED", "2")))
    x = torch.randn(n, t, s, s, 3).to(dtype)
    labels = torch.randint(0, arch.num_classes, (n,))
    mask = (torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float()
    m = _model(cfg, params, dtype, gpu)
    m.set_dropout_mask(mask)
    ls = 1024.0 if dtype == torch.float16 else 1.0
    xg = x.to(gpu)
    pl = m.forward_backward(xg, labels.to(gpu), loss_scale=ls)          # builds the plan
    torch.cuda.synchronize()
    assert any(getattr(B, "a_bwd_rc", False) for B in pl.blocks), "no recomputed-output launch in this plan: nothing to compare"
    alt = record_alternate_backward(m, pl, xg, pw_bwd_rc=False)
    assert not any(alt.info["a_bwd_rc"])
    names_rc = {i for i, (nm, fn, a_) in enumerate(pl.bwd) if nm == "x3d_bn_bwd_finalize_rc"}
    assert names_rc and not any(nm in ("x3d_bn_bwd_finalize_rc", "x3d_pw_bwd_rc_prepare", "x3d_pw_bwd_rc_finish") for nm, _, _ in alt.lst)

    # which gradients can a recomputed-output launch reach?  Walk the blocks in backward order: everything recorded before the
    # first such launch is out of reach; from there on every dx -- and every gradient computed from it -- is downstream.
    reach, seen = {}, False
    for B in reversed(pl.blocks):
        pre = block_prefix(B.spec)
        for k in m.grads:
            if k.startswith(pre + "/"):
                reach[k] = seen
        if getattr(B, "r_bwd_rc", False):
            reach[pre + "/residual/kernel"] = True
            seen = True
        if getattr(B, "a_bwd_rc", False):
            reach[pre + "/bottleneck/a/kernel"] = True
            seen = True
    for k in m.grads:
        reach.setdefault(k, seen if k.startswith("conv1/") else False)      # the stem is last, the head first
    assert any(reach.values()) and not all(reach.values())

    def one_round(tag):
        m._pack_panels()
        pl.zero_buf.zero_()
        pl.run(pl.fwd, 0, pl.grad_scale_slot)
        hip.call("x3d_softmax_xent", pl.logits.data_ptr(), pl.labels.data_ptr(), pl.probs.data_ptr(),
                 pl.loss_rows.data_ptr(), pl.dlogits.data_ptr(), ls / n, n, arch.num_classes)
        snap = pl.zero_buf.clone()
        out, dpool = [], []
        se_blocks = [B for B in reversed(pl.blocks) if B.spec.has_se]

        def run_split(lst, scratch):
            """the list in pieces that end behind each x3d_se_bnb_bwd launch: d(loss)/d(pooled) of every SE block, [N][inner],
            as that launch leaves it at the head of its scratch (x3d_hip.h, x3d_se_bnb_bwd_args.scratch) -- the vector BEFORE
            the SE layers project it onto sums that cancel"""
            idx = [i for i, item in enumerate(lst) if item[0] == "x3d_se_bnb_bwd"]      # one per block (BN_b backward; SE or not)
            assert len(idx) == len(pl.blocks), (len(idx), len(pl.blocks))
            got, start = [], 0
            for i, B in zip(idx, reversed(pl.blocks)):
                if not B.spec.has_se:
                    continue
                pl.run(lst, start, i + 1)
                torch.cuda.synchronize()
                got.append(scratch[:n * B.spec.inner].double().cpu().clone())
                start = i + 1
            pl.run(lst, start)
            return got

        def run_alt():
            alt.extra.zero_()
            return run_split(alt.lst, alt.owned[5])         # (owned: the alternate list's scratch buffers, se_scratch sixth)

        for runner in (lambda: run_split(pl.bwd, pl.se_scratch), run_alt, lambda: run_split(pl.bwd, pl.se_scratch)):
            pl.zero_buf.copy_(snap)
            m.flat_grads.zero_()
            dpool.append(runner())
            torch.cuda.synchronize()
            assert torch.isfinite(m.flat_grads).all()
            out.append({k: g.detach().double().cpu().clone() for k, g in m.grads.items()})
        g1, g0, g1b = out
        lim_same, lim, lim_g, lim_se, lim_dp = RC_DIFF_LIMITS[dtype]
        # d(loss)/d(pooled) of every SE block, the vector the SE layers project (round 6; asked for as the "pre-cancellation"
        # quantity of the SE class).  Measured (profiles/r06_rc_diff_sweep.log, seeds 2 .. 7): it is ITSELF a cancelling sum --
        # over the 50 K .. 200 K points of a plane, of swish-backward terms of either sign -- and sits at 1.3e-2 .. 1.8e-1 in bf16
        # (worst: X3D-L stage-2 blocks) and up to 5.7e-2 even in fp16, i.e. in the class of the projections or beyond it, not at
        # the 3e-2 of a general tensor.  Bounded as its own class (an O(1) error there fails); the tensor-level statements are
        # test_pw_bwd_rc (2e-2 of the maximum per layer) and the full-size checks.
        dp_err = [((a - b).norm() / (b.norm() + 1e-30)).item() for a, b in zip(dpool[0], dpool[1])]
        dp_same = [((a - b).norm() / (b.norm() + 1e-30)).item() for a, b in zip(dpool[0], dpool[2])]
        print(f"rc differential {tag}: d(loss)/d(pooled) of the {len(se_blocks)} SE blocks, worst rc vs stored {max(dp_err):.2e}, rc vs rc again {max(dp_same):.1e}")
        assert max(dp_err) <= lim_dp and max(dp_same) <= 50 * lim_same, (dp_err, dp_same)
        errs = {}
        for B in pl.blocks + [None]:
            ks = [k for k in g1 if (k.startswith(block_prefix(B.spec) + "/") if B is not None else not k.startswith("stages/"))]
            rms = sorted(g0[k].norm().item() / g0[k].numel() ** 0.5 for k in ks)
            floor_rms = 0.05 * rms[len(rms) // 2]           # (cancelling sums: as in the teacher-forced test)
            for k in ks:
                den = max(g0[k].norm().item(), floor_rms * g0[k].numel() ** 0.5) + 1e-30
                errs[k] = ((g1[k] - g0[k]).norm().item() / den, (g1[k] - g1b[k]).norm().item() / den)
        worst = sorted(errs.items(), key=lambda kv: -kv[1][0])[:4]
        cls = lambda k: ("se" if k.endswith(("/se_fc1/bias", "/se_fc2/bias", "/se_fc1/kernel")) else ("gamma" if k.endswith("/gamma") else "general"))
        by_class = {c: max([e for k, (e, _) in errs.items() if reach[k] and cls(k) == c] + [0.0]) for c in ("general", "gamma", "se")}
        print(f"rc differential {tag}:", name, dtype, "worst per class", {c: f"{v:.2e}" for c, v in by_class.items()},
              "worst (rc vs stored, rc vs rc again):", worst, "out of reach:", sum(1 for v in reach.values() if not v), "of", len(reach))
        bad = {}
        for k, (e, e_same) in errs.items():
            limit = lim_same if not reach[k] else {"se": lim_se, "gamma": lim_g, "general": lim}[cls(k)]
            if e > limit or e_same > lim_same * (1 if not reach[k] else 50):
                bad[k] = (e, e_same, limit)
        assert not bad, f"{tag}: recomputed-output backward vs stored-output backward beyond the limits: {bad}"

    one_round("step 1")
    # other weights, same plan: the per-step operands (panel, c0, coefficient tables) must follow
    g = torch.Generator(device="cpu")
    g.manual_seed(11)
    nt = m.n_trainable_flat
    m.flat_params[:nt].mul_((1.0 + 0.25 * torch.randn(nt, generator=g)).to(gpu))
    one_round("step 2 (perturbed weights)")


@pytest.mark.parametrize("dtype", S.HALF_DTYPES)
@pytest.mark.parametrize("off", ["coef_fold", "dw_slab"])
def test_launch_saving_plan_options_do_not_change_the_gradients(gpu, off, dtype):
    """Round 5's two launch-list options -- coef_fold (the BatchNorm-backward finalize derived by its consumers: 37 launches of an
    X3D-M step) and dw_slab (partial weight-gradient slabs added up by x3d_se_bnb_bwd instead of fp32 atomics) -- change WHERE
    numbers are computed, not the numbers: on the same forward state the list recorded with the option off gives the same
    gradients up to the summation order of the remaining fp32 atomics (the headline's planes, X3D-M 1 x 4 x 224^2)."""
    from tests.util import record_alternate_backward
    from x3d_tf_amd import hip
    cfg, arch, params = _setup("M")
    torch.manual_seed(3)
    n, t, s = 1, 4, 224
    x = torch.randn(n, t, s, s, 3).to(dtype).to(gpu)
    labels = torch.randint(0, arch.num_classes, (n,)).to(gpu)
    m = _model(cfg, params, dtype, gpu)
    m.set_dropout_mask((torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float())
    ls = 1024.0 if dtype == torch.float16 else 1.0
    pl = m.forward_backward(x, labels, loss_scale=ls)
    torch.cuda.synchronize()
    names = [i[0] for i in pl.bwd]
    alt = record_alternate_backward(m, pl, x, **{off: False})
    alt_names = [i[0] for i in alt.lst]
    if off == "coef_fold":
        assert names.count("x3d_bn_bwd_finalize") == 0 and alt_names.count("x3d_bn_bwd_finalize") == 37
    else:
        slabbed = lambda lst: sum(1 for i, it in enumerate(lst) if it[0] in ("x3d_pw_bwd", "x3d_pw_wgrad") and pl.structs[(id(lst), i)].dw_slab)
        assert slabbed(pl.bwd) >= 20 and slabbed(alt.lst) == 0      # (stage 4: 21 launches; stage 5 has P % 8 != 0 at T = 4)
    m._pack_panels()
    pl.zero_buf.zero_()
    pl.run(pl.fwd, 0, pl.grad_scale_slot)
    hip.call("x3d_softmax_xent", pl.logits.data_ptr(), pl.labels.data_ptr(), pl.probs.data_ptr(), pl.loss_rows.data_ptr(),
             pl.dlogits.data_ptr(), ls / n, n, arch.num_classes)
    snap = pl.zero_buf.clone()
    out = []
    for runner in (lambda: pl.run(pl.bwd), alt.run):
        pl.zero_buf.copy_(snap)
        m.flat_grads.zero_()
        runner()
        torch.cuda.synchronize()
        assert torch.isfinite(m.flat_grads).all()
        out.append({k: g.detach().double().cpu().clone() for k, g in m.grads.items()})
    g1, g0 = out
    scale = {k: max(v.norm().item(), 1e-30) for k, v in g0.items()}
    med = sorted(scale[k] / g0[k].numel() ** 0.5 for k in g0)[len(g0) // 2]
    errs = {k: (g1[k] - g0[k]).norm().item() / max(scale[k], 0.05 * med * g0[k].numel() ** 0.5) for k in g0}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print("plan option", off, dtype, "worst:", worst)
    assert worst[0][1] < 2e-5, worst


def test_train_step_bf16_end_to_end_sanity(gpu):
    """Free-running bf16 vs the fp32 oracle: only what chaos leaves meaningful -- probabilities within 3e-3
    abs (they are ~2.5e-3 each), cross-entropy within 2 %, all gradients finite and of the right magnitude
    (|g|_2 of the whole gradient within 30 %)."""
    from oracle import x3d_oracle as O
    cfg, arch, params = _setup("XS")
    n, t, s = 4, 4, 96
    torch.manual_seed(2)
    x = torch.randn(n, t, s, s, 3).bfloat16().float()
    labels = torch.randint(0, arch.num_classes, (n,))
    mask = (torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float()
    m = _model(cfg, params, torch.bfloat16, gpu)
    m.set_dropout_mask(mask)
    pl = m.forward_backward(x.to(gpu), labels.to(gpu))
    torch.cuda.synchronize()
    r = O.train_step({k: v.clone() for k, v in params.items()}, x, labels, arch, lr=None, dropout_mask=mask,
                     apply_update=False)
    report("probs", pl.probs, r["probs"], 0, 3e-3)
    report("ce", pl.loss_rows.mean().view(1), r["ce"].view(1), 2e-2, 0)
    assert torch.isfinite(m.flat_grads).all()
    g_dev = torch.cat([m.grads[k].reshape(-1).cpu().double() for k in r["grads"]])
    g_ref = torch.cat([(r["grads"][k].double() - (2 * arch.weight_decay * params[k].double() if m.specs[k].l2 else 0)).reshape(-1)
                       for k in r["grads"]])
    assert abs(g_dev.norm().item() / g_ref.norm().item() - 1.0) < 0.3


def test_summary_and_surface(gpu):
    """The reference's module surface: attribute tree, summary table, training flag semantics."""
    cfg, arch, params = _setup("M")
    m = _model(cfg, params, torch.float32, gpu)
    text = m.summary((16, 224, 224, 3), print_fn=None)
    assert "Total params: 3,795,830" in text and "Trainable params: 3,764,366" in text
    assert "(None, 16, 7, 7, 192)" in text
    blk = m.stages[0].stage[0]
    assert tuple(blk.bottleneck.a.kernel.shape) == (54, 24)
    assert tuple(blk.residual.kernel.shape) == (24, 24) and hasattr(blk, "bn_r")
    assert hasattr(m.stages[0].stage[0].bottleneck, "se_fc1") and not hasattr(m.stages[0].stage[1].bottleneck, "se_fc1")
    assert m.conv1.conv_t.kernel.shape == (24, 5) and m.fc2.bias.shape == (400,)
    # training=True returns per-clip probabilities, no view averaging
    out = m(torch.randn(2, 4, 32, 32, 3, device=gpu), training=True)
    assert tuple(out.shape) == (2, 400)


def test_checkpoint_roundtrip_through_model(gpu, tmp_path):
    """save_weights -> TF-bundle files -> load_weights into a fresh model reproduces the forward bit for bit."""
    cfg, arch, params = _setup("XS", ["TEST.NUM_TEMPORAL_VIEWS", 2])
    m = _model(cfg, params, torch.float32, gpu)
    m.flat_velocity.normal_()
    prefix = str(tmp_path / "ckpt" / "model")
    m.save_weights(prefix)
    from x3d_tf_amd.model import X3D
    m2 = X3D(cfg, dtype=torch.float32, device=gpu, seed=123)
    m2.load_weights(str(tmp_path / "ckpt"))          # directory -> latest_checkpoint
    x = torch.randn(2, 4, 64, 64, 3, device=gpu)
    a = m(x).clone()
    b = m2(x)
    assert torch.equal(a, b)
    for k in m.grads:                                  # momentum slots travel too (alignment gaps excluded)
        o, nel = m._offsets[k], m.params[k].numel()
        assert torch.equal(m.flat_velocity[o:o + nel], m2.flat_velocity[o:o + nel]), k


def test_trainer_checkpoint_resume(gpu, tmp_path):
    """Trainer.save_checkpoint writes the reference's `ckpt-<epoch>` + `checkpoint` state file (utils.py:128-132);
    Trainer.resume finds it like train.py:131-136 and restores weights and momentum bit for bit."""
    from x3d_tf_amd.model import X3D
    from x3d_tf_amd.train import Trainer
    cfg, arch, params = _setup("XS", ["NETWORK.NUM_CLASSES", 7])
    m = X3D(cfg, dtype=torch.float32, device=gpu, seed=5)
    tr = Trainer(m, cfg)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 64, 64, 3, generator=g).to(gpu)
    y = torch.randint(0, 7, (2,), generator=g).to(gpu)
    m.set_dropout_mask(torch.ones(2, arch.fc1_out))
    tr.step(x, y, 0.05)
    prefix = tr.save_checkpoint(str(tmp_path / "run"), epoch=3)
    assert prefix.endswith("ckpt-3") and (tmp_path / "run" / "checkpoint").exists()
    m2 = X3D(cfg, dtype=torch.float32, device=gpu, seed=99)
    tr2 = Trainer(m2, cfg)
    assert tr2.resume(str(tmp_path / "run")) == 3 and tr2.epoch == 3
    assert Trainer(X3D(cfg, dtype=torch.float32, device=gpu, seed=1), cfg).resume(str(tmp_path / "empty")) == 0
    for k in m.grads:                                  # restored state is bit-identical (weights and momentum)
        o, nel = m._offsets[k], m.params[k].numel()
        assert torch.equal(m.params[k], m2.params[k]), k
        assert torch.equal(m.flat_velocity[o:o + nel], m2.flat_velocity[o:o + nel]), k
    m2.set_dropout_mask(torch.ones(2, arch.fc1_out))
    tr.step(x, y, 0.05)
    tr2.step(x, y, 0.05)
    torch.cuda.synchronize()
    for k in m.grads:   # the continued step agrees to fp32 atomic-summation order (weight gradients use fp32 atomics)
        d = (m.params[k] - m2.params[k]).abs().max().item()
        assert d <= 1e-5 * max(1.0, m.params[k].abs().max().item()), (k, d)


def test_trainer_adam_checkpoint_resume_matches_uninterrupted_run(gpu, tmp_path):
    """OPTIMIZER adam (train.py:93-95) through save_checkpoint -> resume: both moments and the step count (`optimizer/iter`,
    Adam's bias correction) are restored, so the step after a resume equals the step of the run that never stopped.
    (Round 2 restored the first moment only, with v = 0 and t = 1: a ~10x first update.)"""
    import x3d_tf_amd as x
    from x3d_tf_amd.model import X3D
    from x3d_tf_amd.train import Trainer
    cfg = x.get_config("XS", ["TRAIN.OPTIMIZER", "adam", "NETWORK.NUM_CLASSES", 7])
    arch = x.build_arch(cfg)
    m = X3D(cfg, dtype=torch.float32, device=gpu, seed=5)
    tr = Trainer(m, cfg)
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(2, 4, 64, 64, 3, generator=g).to(gpu)
    y1 = torch.randint(0, 7, (2,), generator=g).to(gpu)
    m.set_dropout_mask(torch.ones(2, arch.fc1_out))
    for _ in range(3):
        tr.step(x1, y1, 1e-3)
    tr.save_checkpoint(str(tmp_path / "run"), epoch=1)
    m2 = X3D(cfg, dtype=torch.float32, device=gpu, seed=99)
    tr2 = Trainer(m2, cfg)
    assert tr2.resume(str(tmp_path / "run")) == 1 and tr2.opt_step == 3
    for k in m.grads:
        o, nel = m._offsets[k], m.params[k].numel()
        assert torch.equal(m.params[k], m2.params[k]), k
        assert torch.equal(m.flat_velocity[o:o + nel], m2.flat_velocity[o:o + nel]), k
        assert torch.equal(m.flat_second[o:o + nel], m2.flat_second[o:o + nel]), k
    m2.set_dropout_mask(torch.ones(2, arch.fc1_out))
    before = m.flat_params[:m.n_trainable_flat].clone()
    tr.step(x1, y1, 1e-3)
    tr2.step(x1, y1, 1e-3)
    torch.cuda.synchronize()
    step = (m.flat_params[:m.n_trainable_flat] - before).abs().max().item()
    assert step <= 1.5e-3                      # an Adam step is bounded by ~lr; the broken resume moved weights by ~10 lr
    for k in m.grads:   # the continued step agrees to fp32 atomic-summation order (weight gradients use fp32 atomics)
        d = (m.params[k] - m2.params[k]).abs().max().item()
        assert d <= 2e-5 * max(1.0, m.params[k].abs().max().item()), (k, d)
    # a checkpoint of the OTHER optimizer branch: variables restored, slots start from zero (never SGD momentum as Adam's m)
    cfg_s = x.get_config("XS", ["TRAIN.OPTIMIZER", "sgd", "NETWORK.NUM_CLASSES", 7])
    m3 = X3D(cfg_s, dtype=torch.float32, device=gpu, seed=7)
    tr3 = Trainer(m3, cfg_s)
    assert tr3.resume(str(tmp_path / "run")) == 1 and tr3.opt_step == 0
    assert float(m3.flat_velocity.abs().max()) == 0.0
    # ... and WITHOUT a Trainer: a bare model.load_weights() of the Adam bundle installs Adam's moments, an SGD
    # update then starts from zero momentum instead of using Adam's first moment as velocity -- and vice versa
    m4 = X3D(cfg_s, dtype=torch.float32, device=gpu, seed=8)
    m4.load_weights(str(tmp_path / "run"))
    assert m4.slot_kind == "adam" and float(m4.flat_velocity.abs().max()) > 0.0
    m4.flat_grads.zero_()
    w_before = m4.flat_params[:m4.n_trainable_flat].clone()
    m4.apply_sgd(0.1, 0.9)                    # zero gradient + zero (re-initialised) momentum: only the L2 term moves weights
    torch.cuda.synchronize()
    assert m4.slot_kind == "sgd"
    l2 = m4.l2_mask.bool()
    assert torch.equal(m4.flat_params[:m4.n_trainable_flat][~l2], w_before[~l2])
    m5 = X3D(cfg_s, dtype=torch.float32, device=gpu, seed=8)
    m5.load_weights(str(tmp_path / "run"), optimizer="sgd")      # the caller names its branch: nothing of Adam's is installed
    assert m5.slot_kind is None and float(m5.flat_velocity.abs().max()) == 0.0


FULL_SIZE_TRAIN = [          # BASELINE configs at FULL size: variant, clips, T, S, storage, clips per inference sub-plan
    ("M", 64, 16, 224, torch.bfloat16, 8),      # config 3: the plan bench.py times
    ("S", 32, 13, 160, torch.float32, 8),       # config 2: fp32, 13 frames (rows of P % 8 != 0 points in stages 4 / 5)
    ("L", 16, 16, 312, torch.bfloat16, 4),      # config 4: 156 / 78 / 39 / 20 / 10 planes (odd 39 -> 20), 55 blocks
]


@pytest.mark.parametrize("name,n,t,s,dtype,part", FULL_SIZE_TRAIN, ids=[c[0] for c in FULL_SIZE_TRAIN])
def test_full_size_plan_properties(gpu, name, n, t, s, dtype, part):
    """BASELINE configs 3, 2 and 4 at FULL size (the workgroup counts, 32-bit offsets and replica indexing of the real
    plans), checked through size-independent properties, since no CPU oracle finishes these sizes:
      * loss and every gradient finite;
      * the BatchNorm batch statistics the kernels' epilogues accumulated equal an fp64 reduction of the stored tensors;
      * linearity of the explicit backward pass: doubling the upstream gradient (loss_scale = 2) doubles every gradient;
      * per-sample independence in inference mode (moving statistics): the n-clip plan gives the same probabilities as
        n / part plans of `part` clips -- tile / workgroup partitioning across samples does not leak between clips."""
    import x3d_tf_amd as x
    from x3d_tf_amd.model import X3D
    from x3d_tf_amd.params import init_params, randomize_bn_
    cfg = x.get_config(name, ["TEST.NUM_TEMPORAL_VIEWS", 1, "TEST.NUM_SPATIAL_CROPS", 1])
    arch = x.build_arch(cfg)
    m = X3D(cfg, dtype=dtype, device=gpu, seed=0)
    m.load_state_dict(randomize_bn_(init_params(arch, seed=3), seed=4))
    g = torch.Generator(device=gpu)
    g.manual_seed(7)
    clips = torch.randn((n, t, s, s, 3), generator=g, device=gpu).to(dtype)
    labels = torch.randint(0, arch.num_classes, (n,), generator=g, device=gpu)
    mask = (torch.rand((n, arch.fc1_out), generator=g, device=gpu) >= arch.dropout_rate).float()
    m.set_dropout_mask(mask)
    moving0 = m.moving_stats_flat().clone()
    pl = m.forward_backward(clips, labels)
    torch.cuda.synchronize()
    assert torch.isfinite(pl.loss_rows).all() and torch.isfinite(m.flat_grads).all()
    assert 4.0 < pl.loss_rows.mean().item() < 9.0           # ~ln(400) = 5.99 for a random-init classifier
    g1 = m.flat_grads.clone()
    # batch statistics: first two blocks (the widest planes), a mid-network block, the last block
    nb = len(pl.blocks)
    for bi in (0, 1, nb // 2, nb - 1):
        B = pl.blocks[bi]
        for raw, bn in ((B.a_raw, B.bn_a), (B.b_raw, B.bn_b), (B.c_raw, B.bn_c)):
            d = raw.double()
            mean = d.mean((0, 2, 3, 4))
            var = d.var((0, 2, 3, 4), unbiased=False)
            del d
            mi = bn.mi.double()
            scale = max(mean.abs().max().item(), var.sqrt().max().item())
            report(f"block {bi} {bn.prefix} mean", mi[:, 0], mean, 0, 2e-4 * scale)
            report(f"block {bi} {bn.prefix} invstd", mi[:, 1], 1 / torch.sqrt(var + arch.bn_eps), 5e-4, 0)
    # linearity
    m.moving_stats_flat().copy_(moving0)
    pl = m.forward_backward(clips, labels, loss_scale=2.0)
    torch.cuda.synchronize()
    g2 = m.flat_grads
    num = (g2 - 2 * g1).norm().item()
    assert num <= 2e-3 * (2 * g1).norm().item(), f"backward pass not linear in the upstream gradient: {num}"
    # per-sample independence (inference)
    m.moving_stats_flat().copy_(moving0)
    m.release_plans()
    full = m(clips, training=False).clone()
    assert tuple(full.shape) == (n, arch.num_classes) and torch.isfinite(full).all()
    parts = torch.cat([m(clips[i:i + part], training=False).clone() for i in range(0, n, part)], 0)
    report(f"probs {n} vs {n // part}x{part}", full, parts, 0, 1e-6)
    m.release_plans()


def test_full_size_xl_30_view_inference_properties(gpu):
    """BASELINE config 5 at FULL size: X3D-XL, one video = 10 temporal views x 3 spatial crops of 16x312x312, fp16 storage
    (the reference's mixed_float16), inference.  Properties: the output is one finite probability row that sums to 1; the
    30-clip plan equals the mean of three 10-clip plans (view averaging, model.py:123-126, and no leakage between clips
    across the tile / workgroup partitioning of the large plan); a second video in the same batch does not change the
    first one's row."""
    import x3d_tf_amd as x
    from x3d_tf_amd.model import X3D
    from x3d_tf_amd.params import init_params, randomize_bn_
    cfg30 = x.get_config("XL", ["TEST.NUM_TEMPORAL_VIEWS", 10, "TEST.NUM_SPATIAL_CROPS", 3])
    cfg10 = x.get_config("XL", ["TEST.NUM_TEMPORAL_VIEWS", 10, "TEST.NUM_SPATIAL_CROPS", 1])
    arch = x.build_arch(cfg30)
    params = randomize_bn_(init_params(arch, seed=3), seed=4)
    g = torch.Generator(device=gpu)
    g.manual_seed(11)
    clips = torch.randn((60, 16, 312, 312, 3), generator=g, device=gpu).to(torch.float16)
    m30 = X3D(cfg30, dtype=torch.float16, device=gpu, seed=0)
    m30.load_state_dict(params)
    one = m30(clips[:30]).clone()
    assert tuple(one.shape) == (1, arch.num_classes) and torch.isfinite(one).all()
    assert abs(one.sum().item() - 1.0) < 1e-5 and one.min().item() >= 0
    two = m30(clips).clone()                                  # two videos in one batch
    assert tuple(two.shape) == (2, arch.num_classes)
    report("video 0 alone vs in a batch of two", two[:1], one, 0, 1e-6)
    m30.release_plans()
    del m30
    m10 = X3D(cfg10, dtype=torch.float16, device=gpu, seed=0)
    m10.load_state_dict(params)
    parts = torch.cat([m10(clips[i:i + 10]).clone() for i in range(0, 30, 10)], 0)
    report("30-view row vs mean of three 10-view rows", one, parts.mean(0, keepdim=True), 0, 1e-6)
    m10.release_plans()


@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_trainer_fp16_loss_scaling_and_adam(gpu, opt):
    """reference train.py:88-100: SGD(nesterov) / Adam, wrapped in LossScaleOptimizer under mixed_float16.  fp16 storage:
    dynamic loss scale (2^15, halved + step skipped on a non-finite gradient); the update applies the UNSCALED gradient:
    it matches a step taken at loss scale 1024 to fp16 noise, and an fp64 restatement of the optimizer formula exactly."""
    import x3d_tf_amd as x
    from x3d_tf_amd.model import X3D
    from x3d_tf_amd.train import Trainer
    cfg = x.get_config("XS", ["TRAIN.OPTIMIZER", opt, "NETWORK.NUM_CLASSES", 11])
    arch = x.build_arch(cfg)
    gen = torch.Generator().manual_seed(3)
    clips = torch.randn(2, 4, 64, 64, 3, generator=gen).to(gpu)
    labels = torch.randint(0, 11, (2,), generator=gen).to(gpu)
    m = X3D(cfg, dtype=torch.float16, device=gpu, seed=5)
    m.set_dropout_mask(torch.ones(2, arch.fc1_out))
    tr = Trainer(m, cfg)
    assert tr.dynamic_scale and tr.loss_scale == 2.0 ** 15 and tr.optimizer == opt
    w0 = m.flat_params[:m.n_trainable_flat].double().cpu().clone()
    lr = 0.01
    tr.step(clips, labels, lr)
    torch.cuda.synchronize()
    assert tr.skipped_steps == 0 and tr.opt_step == 1 and torch.isfinite(m.flat_params).all()
    # fp64 restatement of the update from the gradient buffer the step left behind (scaled by the loss scale)
    gsc = m.flat_grads.double().cpu() / tr.loss_scale
    gsc = gsc + 2 * arch.weight_decay * w0 * m.l2_mask.double().cpu()
    if opt == "sgd":
        v = -lr * gsc
        want = w0 + cfg.TRAIN.MOMENTUM * v - lr * gsc
    else:
        mm, vv = 0.1 * gsc, 0.001 * gsc * gsc
        want = w0 - lr * (1 - 0.999) ** 0.5 / (1 - 0.9) * mm / (vv.sqrt() + 1e-7)
    got = m.flat_params[:m.n_trainable_flat].double().cpu()
    assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    # an overflowing loss scale: the step is skipped, the scale halved, the weights untouched
    tr.loss_scale = 2.0 ** 40
    before = m.flat_params.clone()
    tr.step(clips, labels, lr)
    torch.cuda.synchronize()
    assert tr.skipped_steps == 1 and tr.loss_scale == 2.0 ** 39 and tr.opt_step == 1
    assert torch.equal(before[:m.n_trainable_flat], m.flat_params[:m.n_trainable_flat])
    with pytest.raises(NotImplementedError):
        Trainer(m, x.get_config("XS", ["TRAIN.OPTIMIZER", "rmsprop"]))
