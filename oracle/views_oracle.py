"""TEST INFRASTRUCTURE ONLY -- CPU (NumPy fp32) restatement of the reference's eval-side view construction.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(x3d_tf_amd.views) never does.

PARITY UNPINNED: the reference has no test or golden vector for this path and TensorFlow is not installable here;
the two TF-internal rules used below are marked [TF-3p].

Follows (reference file:line):
  * temporal looping sampler, eval branch ............ transforms.py:48-65, 70-84
  * short-side resize to TEST_CROP_SIZE, cast back .... transforms.py:112-147 (called with min = max = crop, :215-218)
  * uniform (left/centre/right or top/centre/bottom) crop, ceil offsets .. transforms.py:149-190, :219-225
  * normalise  x/255 - mean, / std  per channel ........ utils.py:42-72, transforms.py:228
  * flattening order [crops][views] -> clips ........... dataloader.py:107-116
Training branch (train_clip below):
  * random start, every rate-th frame, looped video .... transforms.py:31-47
  * random_short_side_resize with a float target ....... transforms.py:112-147
  * tf.image.random_crop (one offset per clip) [TF-3p] . transforms.py:199-203
  * flip_left_right on every training clip ............. transforms.py:205-206, dataloader.py:136
"""
import math

import numpy as np


def temporal_indices(num_video_frames: int, num_frames: int, num_views: int):
    """transforms.py:48-58: sample_rate = max(1, size // T); the video is looped (tf.tile) until T*rate*views
    frames exist and every rate-th one is taken: frame j of the sweep is (j * rate) mod size."""
    size = int(num_video_frames)
    rate = max(1, size // num_frames)
    end = num_frames * rate * num_views
    loops = int(math.ceil(end / size))
    idx = np.tile(np.arange(size), loops)[:end][0:end:rate]
    assert idx.shape[0] == num_frames * num_views
    return idx.reshape(num_views, num_frames)


def resized_hw(height: int, width: int, size: int):
    """transforms.py:129-141.  The arithmetic is float32 as in the reference (tf.float32 casts at :126-127)."""
    h, w, s = np.float32(height), np.float32(width), np.float32(size)
    if (w <= h and w == s) or (h <= w and h == s):
        return height, width
    new_h, new_w = s, s
    if w < h:
        new_h = np.floor((h / w) * s)
    else:
        new_w = np.floor((w / h) * s)
    return int(new_h), int(new_w)


def resize_bilinear_u8(frames_u8, new_h, new_w):
    """tf.image.resize(..., method=bilinear, antialias=False) on uint8 frames [F,H,W,C] -> float32, then cast back
    to uint8 (transforms.py:142-147).  [TF-3p]: TF2 resize uses half-pixel centres: src = (dst + 0.5) * (in / out) - 0.5,
    lower = max(floor(src), 0), upper = min(ceil(src), in - 1), lerp = src - floor(src), all float32; rows are blended
    after columns (top/bottom lerp in x, then y).  The float -> uint8 cast truncates."""
    f, h, w, c = frames_u8.shape
    if (new_h, new_w) == (h, w):
        return frames_u8.copy()

    def weights(out_size, in_size):
        scale = np.float32(in_size) / np.float32(out_size)
        src = (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        fl = np.floor(src)
        lower = np.maximum(fl, 0).astype(np.int64)
        upper = np.minimum(np.ceil(src), in_size - 1).astype(np.int64)
        return lower, upper, (src - fl).astype(np.float32)

    ylo, yhi, yl = weights(new_h, h)
    xlo, xhi, xl = weights(new_w, w)
    x = frames_u8.astype(np.float32)
    tl = x[:, ylo][:, :, xlo]
    tr = x[:, ylo][:, :, xhi]
    bl = x[:, yhi][:, :, xlo]
    br = x[:, yhi][:, :, xhi]
    xl_ = xl[None, None, :, None]
    yl_ = yl[None, :, None, None]
    top = tl + (tr - tl) * xl_
    bot = bl + (br - bl) * xl_
    out = top + (bot - top) * yl_
    return out.astype(np.uint8)   # truncation, values are inside [0, 255]


def crop_offsets(height: int, width: int, size: int, spatial_idx: int):
    """transforms.py:170-186."""
    y = int(math.ceil((height - size) / 2))
    x = int(math.ceil((width - size) / 2))
    if height > width:
        if spatial_idx == 0:
            y = 0
        elif spatial_idx == 2:
            y = height - size
    else:
        if spatial_idx == 0:
            x = 0
        elif spatial_idx == 2:
            x = width - size
    return y, x


def eval_views(video_u8, num_frames, num_views, num_crops, crop_size, mean, std):
    """video_u8 [F,H,W,3] uint8 -> float32 clips [num_crops * num_views, T, crop, crop, 3], crops major
    (dataloader.py:107-116 flattens [crops, views, ...])."""
    idx = temporal_indices(video_u8.shape[0], num_frames, num_views)
    h, w = video_u8.shape[1:3]
    nh, nw = resized_hw(h, w, crop_size)
    mean = np.asarray(mean, np.float32)
    std = np.asarray(std, np.float32)
    out = np.empty((num_crops, num_views, num_frames, crop_size, crop_size, 3), np.float32)
    for v in range(num_views):
        fr = resize_bilinear_u8(video_u8[idx[v]], nh, nw)
        for ci in range(num_crops):
            sidx = (ci % 3) if num_crops > 1 else 1          # transforms.py:223
            y, x = crop_offsets(nh, nw, crop_size, sidx)
            cr = fr[:, y:y + crop_size, x:x + crop_size, :].astype(np.float32)
            out[ci, v] = (cr / np.float32(255) - mean) / std   # utils.py:59-66
    return out.reshape(num_crops * num_views, num_frames, crop_size, crop_size, 3)


# ---- training branch --------------------------------------------------------------------------------------------
def train_temporal_indices(num_video_frames: int, num_frames: int, rate: int, start: int):
    """transforms.py:31-47: end = start + T*rate; the frame list is tiled ceil(end / size) times and sliced
    [start:end:rate], i.e. frame j = (start + j*rate) mod size."""
    size = int(num_video_frames)
    end = start + num_frames * rate
    loops = int(math.ceil(end / size))
    idx = np.tile(np.arange(size), loops)[start:end:rate]
    assert idx.shape[0] == num_frames
    return idx


def train_resized_hw(height: int, width: int, jitter):
    """transforms.py:124-141 with a NON-integer target: new_width = new_height = size (float32); the long side is
    floor((long / short) * size); both are then cast to int32 (:140-141), which truncates the short side to int(size)."""
    h, w, s = np.float32(height), np.float32(width), np.float32(jitter)
    if (w <= h and w == s) or (h <= w and h == s):
        return height, width
    new_h, new_w = s, s
    if w < h:
        new_h = np.floor((h / w) * s)
    else:
        new_w = np.floor((w / h) * s)
    return int(new_h), int(new_w)


def train_clip(video_u8, num_frames, rate, start, jitter, crop_size, y0, x0, flip, mean, std):
    """video_u8 [F,H,W,3] uint8 -> float32 clip [T, crop, crop, 3] (the 4-D tensor of transforms.py:199-209 before the
    expand_dims).  tf.image.random_crop takes ONE offset for the whole [T, H, W, C] tensor [TF-3p]."""
    idx = train_temporal_indices(video_u8.shape[0], num_frames, rate, start)
    h, w = video_u8.shape[1:3]
    nh, nw = train_resized_hw(h, w, jitter)
    fr = resize_bilinear_u8(video_u8[idx], nh, nw)
    cr = fr[:, y0:y0 + crop_size, x0:x0 + crop_size, :]
    if flip:
        cr = cr[:, :, ::-1, :]                                   # tf.image.flip_left_right: reverse the width axis
    mean = np.asarray(mean, np.float32)
    std = np.asarray(std, np.float32)
    return (cr.astype(np.float32) / np.float32(255) - mean) / std
