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

from tests.util import hip_relu_masks, rel_l2, report

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(name, overrides=None):
    import x3d_tf_amd as x
    from x3d_tf_amd.params import init_params, randomize_bn_
    cfg = x.get_config(name, overrides)
    arch = x.build_arch(cfg)
    params = randomize_bn_(init_params(arch, seed=3), seed=4)
    return cfg, arch, params


def _model(cfg, params, dtype, gpu):
    from x3d_tf_amd.model import X3D
    m = X3D(cfg, dtype=dtype, device=gpu, seed=0)
    m.load_state_dict(params)
    return m


def _scaled(name, got, ref, rel):
    """max abs error relative to the tensor's own magnitude"""
    scale = ref.detach().abs().max().item() + 1e-30
    return report(name, got, ref, 0, rel * scale)


@pytest.mark.parametrize("name,views,t,s", [("XS", 10, 4, 160), ("S", 2, 13, 96)])
def test_forward_inference_fp32(gpu, name, views, t, s):
    """BASELINE config 1: X3D-XS, one video = 10 views of 4x160x160, inference mode with view averaging."""
    from oracle import x3d_oracle as O
    cfg, arch, params = _setup(name, ["TEST.NUM_TEMPORAL_VIEWS", views])
    torch.manual_seed(0)
    x = torch.randn(views, t, s, s, 3)
    ref, ref_logits = O.forward(params, x, arch, training=False, return_logits=True)
    m = _model(cfg, params, torch.float32, gpu)
    out = m(x.to(gpu), training=False)
    torch.cuda.synchronize()
    assert out.dtype == torch.float32 and tuple(out.shape) == (1, arch.num_classes)
    pl = m._plans[(views, t, s, s, False)]
    _scaled("logits", pl.logits, ref_logits, 1e-4)
    report("probs", out, ref, 0, 1e-4)
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


def test_inference_batch_must_be_multiple_of_views(gpu):
    cfg, arch, params = _setup("XS")
    m = _model(cfg, params, torch.float32, gpu)
    with pytest.raises(ValueError):
        m(torch.randn(3, 4, 32, 32, 3, device=gpu), training=False)


@pytest.mark.parametrize("name,n,t,s", [
    ("XS", 4, 4, 64), ("S", 2, 13, 64), ("M", 2, 4, 64), ("S", 3, 5, 96),
    ("XS", 2, 4, 78),     # odd extents end to end: 78 -> 39 -> 20 -> 10 -> 5 -> 3 (X3D-L's 39 -> 20 TF-SAME pads, odd stride-2 planes)
    ("M", 2, 16, 112),    # T = 16 and 56 / 28 / 14 / 7 planes: the deep-prefetch depthwise variants (dw_pd.hip) inside the model
])
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
    probs_free = O.forward({k: v.clone() for k, v in params.items()}, x, arch, training=True, dropout_mask=mask,
                           state=st, taps=taps)
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

    # gradients and update with the device's ReLU sign patterns
    ref_p = {k: v.clone() for k, v in params.items()}
    r = O.train_step(ref_p, x, labels, arch, lr=0.05, momentum=0.9, dropout_mask=mask, apply_update=True,
                     relu_masks=hip_relu_masks(pl))
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
    for k in r["grads"]:
        _scaled("updated " + k, m.params[k], ref_p[k], 1e-4)


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


@pytest.mark.parametrize("name,n,t,s", [
    ("S", 3, 5, 96),      # odd point counts: scalar / generic kernel paths
    ("M", 2, 4, 128),     # every P a multiple of 8, 16-byte aligned rows: the fast paths (vector GEMMs, fused pointwise
                          # backward, packed panels, vector depthwise staging) -- the ones the benchmark runs
    ("XL", 2, 4, 64),     # XL widths (72/162/306/630...: off the 32-grid, K > 432), 55 blocks, SE parity across stages
])
def test_train_step_bf16_block_by_block(gpu, name, n, t, s):
    """bf16 activation storage (fp32 arithmetic), checked with TEACHER FORCING: every residual block of the
    device run is replayed on the oracle from the device's own stored block input (forward) and the device's
    own stored upstream gradient (backward), with the oracle rounding to bf16 at the tensors the device
    stores (oracle Storage).  End-to-end comparison is meaningless in bf16: a random-init BN network amplifies
    perturbations ~600x (measured in fp32), and bf16 rounding re-injects any 1e-7 difference as a 4e-3 one, so
    two exact implementations decorrelate to O(30 %) on gradients.  Per block the error is one bf16 ulp class:
    stated tolerance 2 % of each tensor's max for activations/gradients; weight gradients (whose GEMM operands
    are rounded to bf16 for the matrix cores) 6 % relative L2 worst case, 1.5 % median."""
    from oracle import x3d_oracle as O
    cfg, arch, params = _setup(name)
    torch.manual_seed(2)
    x = torch.randn(n, t, s, s, 3).bfloat16().float()
    labels = torch.randint(0, arch.num_classes, (n,))
    mask = (torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float()
    m = _model(cfg, params, torch.bfloat16, gpu)
    m.set_dropout_mask(mask)
    pl = m.forward_backward(x.to(gpu), labels.to(gpu))        # builds the plan, full step
    torch.cuda.synchronize()
    assert torch.isfinite(pl.loss_rows).all() and torch.isfinite(m.flat_grads).all()
    masks = hip_relu_masks(pl)
    st = O.Storage(torch.bfloat16)
    # replay the backward block by block, capturing each block's upstream gradient before it is overwritten
    m.flat_grads.zero_()
    pl.zero_buf.zero_()
    pl.run(pl.fwd, 0, pl.grad_scale_slot)
    from x3d_tf_amd import hip
    hip.call("x3d_softmax_xent", pl.logits.data_ptr(), pl.labels.data_ptr(), pl.probs.data_ptr(),
             pl.loss_rows.data_ptr(), pl.dlogits.data_ptr(), 1.0 / n, n, arch.num_classes)
    pos = 0
    worst = dict(y=0.0, dx=0.0, dw=0.0)
    errs = {}
    for B in reversed(pl.blocks):
        pl.run(pl.bwd, pos, B.bwd_start)
        torch.cuda.synchronize()
        dy = B.dy_view.float().cpu().clone()
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
        worst["y"] = max(worst["y"], _scaled(pre + "/out", B.y, y_ref, 2e-2))
        grads = torch.autograd.grad(y_ref, [xin] + [leaf[k] for k in names], grad_outputs=dy)
        dx_ref = grads[0]
        if B.spec.has_shortcut_conv or True:
            _scaled(pre + "/dx", dx_dev, dx_ref, 2e-2)
        for k, g_ref in zip(names, grads[1:]):
            e = rel_l2(m.grads[k], g_ref)
            worst["dw"] = max(worst["dw"], e)
            errs[k] = e
    print("bf16 teacher-forced worst:", worst, sorted(errs.items(), key=lambda kv: -kv[1])[:4])
    bad = {k: e for k, e in errs.items() if e > 6e-2}
    assert not bad, f"relative L2 error beyond 6 % (teacher-forced, bf16): {bad}"
    assert sorted(errs.values())[len(errs) // 2] < 1.5e-2        # median


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
