"""Eval-side view construction on the GPU: what the reference's eval input pipeline does to one
decoded video before `model(clips, training=False)` -- temporal looping sampler (transforms.py:48-65), short-side
resize to TEST_CROP_SIZE with the cast back to uint8 (:112-147), uniform crop (:149-190), normalisation
(utils.py:42-72) and the [crops][views] clip order the model's view averaging expects (dataloader.py:107-116,
model.py:123-127).  One HIP launch (`x3d_eval_views`); there is no CPU path.

Train-side clip construction (the device half of the training input pipeline -- decoding stays on the
host): `make_train_clip` / `make_train_batch` do what `TemporalTransforms` / `SpatialTransforms` do to one decoded
video in training mode (transforms.py:31-47, 112-147, 199-206, utils.py:42-72), one HIP launch per clip
(`x3d_train_clip`)."""
import ctypes

import torch

from . import hip


def num_views(cfg) -> int:
    return int(cfg.TEST.NUM_TEMPORAL_VIEWS) * int(cfg.TEST.NUM_SPATIAL_CROPS)


def make_eval_views(video_u8: torch.Tensor, cfg, dtype=torch.float32, out: torch.Tensor = None) -> torch.Tensor:
    """video_u8: decoded video [F, H, W, 3] uint8 on the GPU (contiguous).
    Returns clips [crops * views, T, S, S, 3] (channels-last) with T = cfg.DATA.TEMP_DURATION,
    S = cfg.DATA.TEST_CROP_SIZE, views = cfg.TEST.NUM_TEMPORAL_VIEWS, crops = cfg.TEST.NUM_SPATIAL_CROPS."""
    if not video_u8.is_cuda or video_u8.dtype != torch.uint8 or not video_u8.is_contiguous():
        raise hip.X3DHipError("make_eval_views needs a contiguous uint8 GPU tensor [F, H, W, 3] (no CPU fallback)")
    if video_u8.dim() != 4 or video_u8.shape[-1] != int(cfg.DATA.NUM_INPUT_CHANNELS) or video_u8.shape[-1] != 3:
        raise ValueError(f"expected [F, H, W, 3], got {tuple(video_u8.shape)}")
    f, h, w, _ = video_u8.shape
    t, s = int(cfg.DATA.TEMP_DURATION), int(cfg.DATA.TEST_CROP_SIZE)
    v, c = int(cfg.TEST.NUM_TEMPORAL_VIEWS), int(cfg.TEST.NUM_SPATIAL_CROPS)
    if out is None:
        out = torch.empty((c * v, t, s, s, 3), dtype=dtype, device=video_u8.device)
    mean = (hip._f * 3)(*[float(m) for m in cfg.DATA.MEAN])
    std = (hip._f * 3)(*[float(m) for m in cfg.DATA.STD])
    a = hip.EvalViewsArgs(video_u8.data_ptr(), out.data_ptr(), f, h, w, t, v, c, s, mean, std, hip.dtype_code(out.dtype))
    hip.call_struct("x3d_eval_views", a)
    return out


# ---- training clips ---------------------------------------------------------------------------------------------
def train_resized_hw(height: int, width: int, jitter: float):
    """Extents after random_short_side_resize with target `jitter` (transforms.py:126-141); computed by the library."""
    nh, nw = ctypes.c_int(0), ctypes.c_int(0)
    hip.check(hip.load().x3d_train_resized_hw(int(height), int(width), float(jitter), ctypes.byref(nh), ctypes.byref(nw)),
              "x3d_train_resized_hw")
    return nh.value, nw.value


def draw_train_params(num_video_frames: int, height: int, width: int, cfg, generator: torch.Generator = None) -> dict:
    """The random draws of one training clip: start ~ U{0..F-1} (transforms.py:33), jitter ~ U[min, max) float32
    (:124), crop offsets uniform over the valid positions of the resized frame (tf.image.random_crop, :199-203).
    Drawn from a torch CPU generator: TF's random streams are not reproduced [TF-3p]."""
    lo, hi = (float(v) for v in cfg.DATA.TRAIN_JITTER_SCALES)
    crop = int(cfg.DATA.TRAIN_CROP_SIZE)
    start = int(torch.randint(0, int(num_video_frames), (1,), generator=generator))
    u = torch.rand((), generator=generator, dtype=torch.float32)
    jitter = float((torch.tensor(lo, dtype=torch.float32) + u * torch.tensor(hi - lo, dtype=torch.float32)).item())
    nh, nw = train_resized_hw(height, width, jitter)
    if nh < crop or nw < crop:
        raise ValueError(f"resized frame {nh}x{nw} is smaller than TRAIN_CROP_SIZE {crop}")
    y0 = int(torch.randint(0, nh - crop + 1, (1,), generator=generator))
    x0 = int(torch.randint(0, nw - crop + 1, (1,), generator=generator))
    return dict(start=start, jitter=jitter, y0=y0, x0=x0, flip=True)   # flip: every training clip (transforms.py:205-206)


def make_train_clip(video_u8: torch.Tensor, cfg, params: dict = None, generator: torch.Generator = None,
                    dtype=torch.float32, out: torch.Tensor = None) -> torch.Tensor:
    """video_u8: decoded video [F, H, W, 3] uint8 on the GPU (contiguous).  Returns one clip [T, S, S, 3]
    (channels-last) with T = cfg.DATA.TEMP_DURATION, S = cfg.DATA.TRAIN_CROP_SIZE, every cfg.DATA.FRAME_RATE-th frame
    from a random start, the video looped.  `params` (see draw_train_params) fixes the random draws."""
    if not video_u8.is_cuda or video_u8.dtype != torch.uint8 or not video_u8.is_contiguous():
        raise hip.X3DHipError("make_train_clip needs a contiguous uint8 GPU tensor [F, H, W, 3] (no CPU fallback)")
    if video_u8.dim() != 4 or video_u8.shape[-1] != 3:
        raise ValueError(f"expected [F, H, W, 3], got {tuple(video_u8.shape)}")
    f, h, w, _ = video_u8.shape
    if params is None:
        params = draw_train_params(f, h, w, cfg, generator)
    t, s = int(cfg.DATA.TEMP_DURATION), int(cfg.DATA.TRAIN_CROP_SIZE)
    if out is None:
        out = torch.empty((t, s, s, 3), dtype=dtype, device=video_u8.device)
    mean = (hip._f * 3)(*[float(m) for m in cfg.DATA.MEAN])
    std = (hip._f * 3)(*[float(m) for m in cfg.DATA.STD])
    a = hip.TrainClipArgs(video_u8.data_ptr(), out.data_ptr(), f, h, w, t, int(cfg.DATA.FRAME_RATE), int(params["start"]),
                          float(params["jitter"]), s, int(params["y0"]), int(params["x0"]), 1 if params.get("flip", True) else 0,
                          mean, std, hip.dtype_code(out.dtype))
    hip.call_struct("x3d_train_clip", a)
    return out


def make_train_batch(videos, cfg, generator: torch.Generator = None, dtype=torch.float32) -> torch.Tensor:
    """One augmented clip per decoded video -> [B, T, S, S, 3], the tensor `Trainer.step` takes (dataloader.py:96-104)."""
    t, s = int(cfg.DATA.TEMP_DURATION), int(cfg.DATA.TRAIN_CROP_SIZE)
    if not videos:
        raise ValueError("make_train_batch: empty batch")
    out = torch.empty((len(videos), t, s, s, 3), dtype=dtype, device=videos[0].device)
    for i, v in enumerate(videos):
        make_train_clip(v, cfg, generator=generator, dtype=dtype, out=out[i])
    return out
