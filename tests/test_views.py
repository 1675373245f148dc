"""Eval-side view construction: oracle restatement checks on CPU, HIP kernel vs oracle on GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import views_oracle as V  # noqa: E402


# ---- oracle (CPU): hand-checkable properties of the reference semantics --------------------------
def test_temporal_indices_loop_and_rate():
    # 50 frames, T=16, 10 views (X3D-M eval): rate = 50 // 16 = 3, contiguous sweep with wrap-around
    idx = V.temporal_indices(50, 16, 10)
    assert idx.shape == (10, 16)
    flat = idx.reshape(-1)
    assert np.array_equal(flat, (np.arange(160) * 3) % 50)
    # fewer frames than T: rate 1, the video loops
    idx = V.temporal_indices(5, 4, 3)
    assert np.array_equal(idx.reshape(-1), np.arange(12) % 5)


def test_resized_hw_and_crop_offsets():
    assert V.resized_hw(240, 320, 224) == (224, 298)        # floor(320/240*224) = 298
    assert V.resized_hw(320, 240, 224) == (298, 224)
    assert V.resized_hw(224, 300, 224) == (224, 300)        # short side already there: unchanged
    # ceil offsets (transforms.py:170-171): (299-224)/2 = 37.5 -> 38
    assert V.crop_offsets(224, 299, 224, 1) == (0, 38)
    assert V.crop_offsets(224, 298, 224, 0) == (0, 0) and V.crop_offsets(224, 298, 224, 2) == (0, 74)
    assert V.crop_offsets(298, 224, 224, 0) == (0, 0) and V.crop_offsets(298, 224, 224, 2) == (74, 0)


def test_resize_identity_constant_and_upscale():
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (2, 6, 8, 3)).astype(np.uint8)
    assert np.array_equal(V.resize_bilinear_u8(x, 6, 8), x)
    c = np.full((1, 5, 7, 3), 200, np.uint8)
    assert np.array_equal(V.resize_bilinear_u8(c, 9, 13), np.full((1, 9, 13, 3), 200, np.uint8))
    # 2x upscale of a 2-pixel ramp with half-pixel centres: src = (d + 0.5) / 2 - 0.5 -> [-0.25, .25, .75, 1.25]
    r = np.array([[[[0], [100]]]], np.uint8).repeat(3, axis=3)
    out = V.resize_bilinear_u8(r, 1, 4)[0, 0, :, 0]
    assert out.tolist() == [0, 25, 75, 100]


def test_eval_views_order_and_normalisation():
    rng = np.random.default_rng(1)
    vid = rng.integers(0, 256, (9, 20, 28, 3)).astype(np.uint8)
    mean, std = [0.45, 0.40, 0.35], [0.2, 0.25, 0.3]
    out = V.eval_views(vid, num_frames=4, num_views=2, num_crops=3, crop_size=16, mean=mean, std=std)
    assert out.shape == (6, 4, 16, 16, 3)
    # clip 0 = crop 0 (left), view 0; clip 3 = crop 1 (centre), view 1
    idx = V.temporal_indices(9, 4, 2)
    nh, nw = V.resized_hw(20, 28, 16)
    fr = V.resize_bilinear_u8(vid[idx[1]], nh, nw)
    y, x = V.crop_offsets(nh, nw, 16, 1)
    ref = (fr[:, y:y + 16, x:x + 16].astype(np.float32) / np.float32(255) - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    assert np.array_equal(out[3], ref)


def test_train_temporal_indices_and_resize():
    # 30 frames, T=4, rate 5 from frame 22: 22, 27, 32 -> 2, 37 -> 7 (the video loops)
    assert V.train_temporal_indices(30, 4, 5, 22).tolist() == [22, 27, 2, 7]
    assert V.train_temporal_indices(3, 5, 2, 1).tolist() == [1, 0, 2, 1, 0]
    # float target 273.6 on 240x320: short side int(273.6) = 273, long side floor(320/240*273.6) = 364
    assert V.train_resized_hw(240, 320, 273.6) == (273, 364)
    assert V.train_resized_hw(320, 240, 273.6) == (364, 273)
    assert V.train_resized_hw(256, 300, 256.0) == (256, 300)   # short side already there


def test_train_clip_crop_flip_normalise():
    rng = np.random.default_rng(2)
    vid = rng.integers(0, 256, (11, 24, 30, 3)).astype(np.uint8)
    mean, std = [0.45, 0.40, 0.35], [0.2, 0.25, 0.3]
    a = V.train_clip(vid, 4, 2, 9, 26.5, 16, 3, 5, False, mean, std)
    b = V.train_clip(vid, 4, 2, 9, 26.5, 16, 3, 5, True, mean, std)
    assert a.shape == (4, 16, 16, 3) and np.array_equal(a[:, :, ::-1], b)
    nh, nw = V.train_resized_hw(24, 30, 26.5)
    fr = V.resize_bilinear_u8(vid[[9, 0, 2, 4]], nh, nw)        # (9 + 2j) mod 11
    ref = (fr[:, 3:19, 5:21].astype(np.float32) / np.float32(255) - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    assert np.array_equal(a, ref)


# ---- HIP kernel (GPU) ----------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # F, H, W, T, views, crops, size
    (50, 48, 64, 16, 10, 1, 32),      # landscape, centre crop, looping sampler with rate 3
    (7, 64, 48, 4, 3, 3, 32),         # portrait, 3 crops (top/centre/bottom), fewer frames than T*views
    (12, 32, 45, 4, 2, 3, 32),        # short side already at the crop size: no resize, left/centre/right
    (5, 37, 53, 8, 2, 2, 24),         # odd extents, video shorter than T
])
def test_eval_views_kernel_matches_oracle(gpu, case):
    import x3d_tf_amd as x3d
    from x3d_tf_amd.views import make_eval_views
    f, h, w, t, views, crops, size = case
    cfg = x3d.get_config("XS", ["DATA.TEMP_DURATION", t, "DATA.TEST_CROP_SIZE", size, "TEST.NUM_TEMPORAL_VIEWS", views,
                                "TEST.NUM_SPATIAL_CROPS", crops])
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    vid = rng.integers(0, 256, (f, h, w, 3)).astype(np.uint8)
    ref = V.eval_views(vid, t, views, crops, size, cfg.DATA.MEAN, cfg.DATA.STD)
    out = make_eval_views(torch.from_numpy(vid).to(gpu), cfg, dtype=torch.float32)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    # integer stage (resize + truncation) bit-exact: recover the uint8 pixel from the normalised value
    mean, std = np.asarray(cfg.DATA.MEAN, np.float32), np.asarray(cfg.DATA.STD, np.float32)
    px_ref = np.rint((ref * std + mean) * 255).astype(np.int32)
    px_got = np.rint((got * std + mean) * 255).astype(np.int32)
    assert np.array_equal(px_ref, px_got)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())      # and the float normalisation too
    ob = make_eval_views(torch.from_numpy(vid).to(gpu), cfg, dtype=torch.bfloat16)
    assert torch.equal(ob.cpu(), torch.from_numpy(ref).bfloat16())


@pytest.mark.gpu
def test_evaluate_driver(gpu):
    """views -> model(training=False) -> Keras-style metrics; checked against the same numbers computed by hand."""
    import x3d_tf_amd as x3d
    from x3d_tf_amd.evaluate import evaluate
    from x3d_tf_amd.views import make_eval_views
    cfg = x3d.get_config("XS", ["DATA.TEMP_DURATION", 4, "DATA.TEST_CROP_SIZE", 32, "TEST.NUM_TEMPORAL_VIEWS", 2,
                                "TEST.NUM_SPATIAL_CROPS", 3, "TEST.BATCH_SIZE", 2, "NETWORK.NUM_CLASSES", 10])
    model = x3d.X3D(cfg, dtype=torch.float32, device=gpu, seed=3)
    g = torch.Generator().manual_seed(0)
    vids = [(torch.randint(0, 256, (6 + i, 40, 52, 3), generator=g, dtype=torch.uint8).to(gpu), i % 10) for i in range(3)]
    res = evaluate(model, cfg, vids)
    probs = torch.cat([model(make_eval_views(v, cfg), training=False).clone() for v, _ in vids], 0).cpu()
    labels = torch.tensor([l for _, l in vids])
    # what Keras' model.evaluate reports as `loss`: CE from probabilities (clip, -log q_y + log sum q) + the L2 term
    q = probs.double().clamp(1e-7, 1 - 1e-7)
    loss = float((-q.gather(1, labels[:, None]).squeeze(1).log() + q.sum(1).log()).mean()) + float(model.regularization_loss().item())
    acc = float((probs.argmax(1) == labels).float().mean())
    assert res["videos"] == 3 and abs(res["loss"] - loss) < 1e-5 and abs(res["acc"] - acc) < 1e-6
    assert res["top_5_acc"] >= res["acc"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # F, H, W, T, rate, crop, jitter_lo, jitter_hi
    (40, 48, 64, 8, 5, 32, 36.0, 44.0),      # landscape, looping sampler
    (9, 64, 48, 4, 3, 32, 34.0, 40.0),       # portrait, fewer frames than T * rate
    (6, 37, 53, 4, 1, 24, 24.0, 30.0),       # odd extents
])
def test_train_clip_kernel_matches_oracle(gpu, case):
    import x3d_tf_amd as x3d
    from x3d_tf_amd.views import draw_train_params, make_train_batch, make_train_clip, train_resized_hw
    f, h, w, t, rate, crop, jlo, jhi = case
    cfg = x3d.get_config("XS", ["DATA.TEMP_DURATION", t, "DATA.FRAME_RATE", rate, "DATA.TRAIN_CROP_SIZE", crop,
                                "DATA.TRAIN_JITTER_SCALES", [jlo, jhi]])
    rng = np.random.default_rng(f * 1000 + h)
    vid = rng.integers(0, 256, (f, h, w, 3)).astype(np.uint8)
    dvid = torch.from_numpy(vid).to(gpu)
    g = torch.Generator().manual_seed(f)
    for _ in range(4):
        p = draw_train_params(f, h, w, cfg, g)
        assert 0 <= p["start"] < f and jlo <= p["jitter"] < jhi and p["flip"] is True
        assert train_resized_hw(h, w, p["jitter"]) == V.train_resized_hw(h, w, p["jitter"])
        ref = V.train_clip(vid, t, rate, p["start"], p["jitter"], crop, p["y0"], p["x0"], p["flip"], cfg.DATA.MEAN, cfg.DATA.STD)
        got = make_train_clip(dvid, cfg, params=p)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), ref), float(np.abs(got.cpu().numpy() - ref).max())
        gb = make_train_clip(dvid, cfg, params=p, dtype=torch.bfloat16)
        assert torch.equal(gb.cpu(), torch.from_numpy(ref).bfloat16())
    # no flip, and the batch helper: same generator state -> same clips
    p = dict(draw_train_params(f, h, w, cfg, g), flip=False)
    ref = V.train_clip(vid, t, rate, p["start"], p["jitter"], crop, p["y0"], p["x0"], False, cfg.DATA.MEAN, cfg.DATA.STD)
    assert np.array_equal(make_train_clip(dvid, cfg, params=p).cpu().numpy(), ref)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    batch = make_train_batch([dvid, dvid], cfg, generator=g1)
    assert tuple(batch.shape) == (2, t, crop, crop, 3)
    for i in range(2):
        assert torch.equal(batch[i], make_train_clip(dvid, cfg, generator=g2))
    # bad offsets are refused by the library, not clamped
    with pytest.raises(x3d.hip.X3DHipError):
        make_train_clip(dvid, cfg, params=dict(p, y0=10 ** 6))
