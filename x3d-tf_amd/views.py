"""Eval-side view construction on the GPU: what the reference's eval input pipeline does to one
decoded video before `model(clips, training=False)` -- temporal looping sampler (transforms.py:48-65), short-side
resize to TEST_CROP_SIZE with the cast back to uint8 (:112-147), uniform crop (:149-190), normalisation
(utils.py:42-72) and the [crops][views] clip order the model's view averaging expects (dataloader.py:107-116,
model.py:123-127).  One HIP launch (`x3d_eval_views`); there is no CPU path."""
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
