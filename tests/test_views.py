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
    loss = float((-probs.gather(1, labels[:, None]).clamp(1e-7, 1 - 1e-7).log()).mean())
    acc = float((probs.argmax(1) == labels).float().mean())
    assert res["videos"] == 3 and abs(res["loss"] - loss) < 1e-5 and abs(res["acc"] - acc) < 1e-6
    assert res["top_5_acc"] >= res["acc"]
