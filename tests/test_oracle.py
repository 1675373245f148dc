"""The oracle against (a) its committed golden vectors and (b) an independent naive NumPy restatement of the
conv primitives on tiny shapes (loops over taps, explicit TF-SAME padding), so a slip in the torch
formulation cannot hide."""
import json
import os

import numpy as np
import pytest
import torch

import x3d_tf_amd as x
from x3d_tf_amd.params import init_params, randomize_bn_
from oracle import x3d_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _np_depthwise(xn, wn, stride):
    n, c, t, h, w = xn.shape
    ho, wo = -(-h // stride), -(-w // stride)
    ph = max((ho - 1) * stride + 3 - h, 0) // 2
    pw = max((wo - 1) * stride + 3 - w, 0) // 2
    out = np.zeros((n, c, t, ho, wo))
    for kt in range(3):
        for kh in range(3):
            for kw in range(3):
                for to in range(t):
                    ti = to + kt - 1
                    if not 0 <= ti < t:
                        continue
                    for i in range(ho):
                        hi = i * stride + kh - ph
                        if not 0 <= hi < h:
                            continue
                        for j in range(wo):
                            wi = j * stride + kw - pw
                            if 0 <= wi < w:
                                out[:, :, to, i, j] += xn[:, :, ti, hi, wi] * wn[None, :, kt, kh, kw]
    return out


@pytest.mark.parametrize("h,w,stride", [(8, 8, 1), (8, 8, 2), (7, 9, 2), (5, 5, 1), (39, 4, 2)])
def test_depthwise_against_naive_numpy(h, w, stride):
    g = torch.Generator().manual_seed(0)
    xt = torch.randn(2, 3, 4, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(3, 3, 3, 3, generator=g, dtype=torch.float64)
    ref = _np_depthwise(xt.numpy(), wt.numpy(), stride)
    got = O.depthwise3x3x3(xt, wt, stride).numpy()
    assert got.shape == ref.shape and np.allclose(got, ref, atol=1e-12)


def test_pointwise_bn_stem_against_numpy():
    g = torch.Generator().manual_seed(1)
    xt = torch.randn(2, 5, 3, 6, 7, generator=g, dtype=torch.float64)
    wt = torch.randn(4, 5, generator=g, dtype=torch.float64)
    ref = np.einsum("oc,ncthw->nothw", wt.numpy(), xt.numpy()[:, :, :, ::2, ::2])
    assert np.allclose(O.pointwise(xt, wt, 2).numpy(), ref)
    # Keras BN, training mode: biased variance, momentum convention moving = 0.9*moving + 0.1*batch
    p = {"bn/gamma": torch.tensor([1.5, 0.5, 2.0, 1.0, 0.1], dtype=torch.float64), "bn/beta": torch.arange(5, dtype=torch.float64),
         "bn/moving_mean": torch.zeros(5, dtype=torch.float64), "bn/moving_variance": torch.ones(5, dtype=torch.float64)}
    st = O.BNState()
    y = O.batch_norm(xt, p, "bn", True, 1e-5, 0.9, st).numpy()
    xn = xt.numpy()
    mu, var = xn.mean((0, 2, 3, 4)), xn.var((0, 2, 3, 4))
    ref = (xn - mu[None, :, None, None, None]) / np.sqrt(var + 1e-5)[None, :, None, None, None] \
        * p["bn/gamma"].numpy()[None, :, None, None, None] + p["bn/beta"].numpy()[None, :, None, None, None]
    assert np.allclose(y, ref)
    m = xn.size // 5
    assert np.allclose(st.new_moving["bn/moving_mean"].numpy(), 0.1 * mu)
    assert np.allclose(st.new_moving["bn/moving_variance"].numpy(), 0.9 + 0.1 * var * m / (m - 1))
    # stem: symmetric (1,1) spatial pad + stride 2, then 5-tap temporal conv with (2,2) pad, no BN in between
    cfg = x.get_config("XS")
    arch = x.build_arch(cfg)
    prm = {k: v.double() for k, v in init_params(arch, 0).items()}
    xin = torch.randn(1, 3, 3, 5, 6, generator=g, dtype=torch.float64)
    y = O.stem(xin, prm, arch, False, None).numpy()
    ws, wtt = prm["conv1/conv_s/kernel"].numpy(), prm["conv1/conv_t/kernel"].numpy()
    xp = np.pad(xin.numpy(), ((0, 0), (0, 0), (0, 0), (1, 1), (1, 1)))
    s = np.zeros((1, 24, 3, 3, 3))
    for i in range(3):
        for j in range(3):
            patch = xp[:, :, :, 2 * i:2 * i + 3, 2 * j:2 * j + 3]
            s[:, :, :, i, j] = np.einsum("ocab,nctab->not", ws, patch)
    sp = np.pad(s, ((0, 0), (0, 0), (2, 2), (0, 0), (0, 0)))
    tt = sum(sp[:, :, k:k + 3] * wtt[None, :, k, None, None, None] for k in range(5))
    inv = 1 / np.sqrt(prm["conv1/bn/moving_variance"].numpy() + 1e-5)
    ref = np.maximum((tt - prm["conv1/bn/moving_mean"].numpy()[None, :, None, None, None]) * inv[None, :, None, None, None], 0)
    assert np.allclose(y, ref, atol=1e-10)


def test_loss_and_optimizer_rules():
    cfg = x.get_config("XS")
    arch = x.build_arch(cfg)
    probs = torch.tensor([[0.7, 0.2, 0.1], [1e-9, 1 - 1e-9, 0.0]])
    labels = torch.tensor([0, 0])
    p = {"a/kernel": torch.ones(2, 2), "b/se_fc1/kernel": torch.ones(3), "c/bias": torch.ones(4)}
    loss, ce, reg = O.loss_fn(probs, labels, p, arch)
    q0 = torch.tensor([0.7, 0.2, 0.1])
    q1 = torch.tensor([1e-7, 1 - 1e-7, 1e-7])
    ref = ((-torch.log(q0[0]) + torch.log(q0.sum())) + (-torch.log(q1[0]) + torch.log(q1.sum()))) / 2
    assert ce.item() == pytest.approx(ref.item(), rel=1e-6)
    assert reg.item() == pytest.approx(5e-5 * 4)            # only a/kernel: se_fc1 and biases carry no L2
    w, v, gr = {"w": torch.tensor([1.0])}, {"w": torch.tensor([0.5])}, {"w": torch.tensor([2.0])}
    O.sgd_nesterov_(w, gr, v, 0.1, 0.9)
    assert v["w"].item() == pytest.approx(0.9 * 0.5 - 0.2) and w["w"].item() == pytest.approx(1 + 0.9 * 0.25 - 0.2)


def test_oracle_reproduces_committed_golden_vectors():
    gold = json.load(open(os.path.join(GOLDEN, "oracle_xs_forward.json")))
    cfg = x.get_config(gold["config"])
    arch = x.build_arch(cfg)
    p = randomize_bn_(init_params(arch, seed=gold["param_seed"]), seed=gold["bn_seed"])
    torch.manual_seed(gold["input_seed"])
    xin = torch.randn(*gold["input_shape"])
    probs, logits = O.forward(p, xin, arch, training=False, return_logits=True)
    assert torch.allclose(logits[0], torch.tensor(gold["logits_view0"]), rtol=1e-4, atol=1e-4)
    assert torch.allclose(probs[0], torch.tensor(gold["probs"]), atol=1e-6)
    with pytest.raises(ValueError):
        O.forward(p, xin[:3], arch, training=False)       # not a multiple of views*crops (model.py:125)


def test_storage_emulation_is_identity_in_fp32_and_rounds_in_bf16():
    st32, st16 = O.Storage(None), O.Storage(torch.bfloat16)
    t = torch.randn(100, requires_grad=True)
    assert st32.act(t) is t and st32.grad(t) is t
    y = st16.act(t)
    assert torch.equal(y, t.detach().bfloat16().float())
    (g,) = torch.autograd.grad((st16.grad(t) * torch.linspace(0, 1, 100)).sum(), [t])
    assert torch.equal(g, torch.linspace(0, 1, 100).bfloat16().float())
