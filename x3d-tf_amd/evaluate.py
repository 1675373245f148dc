"""Evaluation driver: the part of the reference's eval.py (:75-89) that runs after the model is built --
`model.load_weights(latest_checkpoint).expect_partial()`, then `model.evaluate(...)` with
SparseCategoricalCrossentropy, SparseCategoricalAccuracy ('acc') and SparseTopKCategoricalAccuracy(k=5)
('top_5_acc') (eval.py:48-66).  Videos come as decoded uint8 tensors; the views are built on the GPU (views.py)."""
from typing import Dict, Iterable, Tuple

import torch

from .views import make_eval_views, num_views


class Metrics:
    """Running means in the Keras sense: per-video loss / hits averaged over the videos seen so far."""

    def __init__(self):
        self.n = 0
        self.loss = 0.0
        self.top1 = 0
        self.top5 = 0

    def update(self, probs: torch.Tensor, labels: torch.Tensor):
        """probs [videos, classes] fp32 (already view-averaged by the model); labels [videos]."""
        labels = labels.to(probs.device).long()
        p = probs.gather(1, labels[:, None]).squeeze(1).clamp(1e-7, 1.0 - 1e-7)   # Keras CE from probabilities
        self.loss += float((-p.log()).sum())
        top5 = probs.topk(min(5, probs.shape[1]), dim=1).indices
        self.top1 += int((top5[:, 0] == labels).sum())
        self.top5 += int((top5 == labels[:, None]).any(dim=1).sum())
        self.n += int(labels.numel())

    def result(self) -> Dict[str, float]:
        n = max(self.n, 1)
        return {"loss": self.loss / n, "acc": self.top1 / n, "top_5_acc": self.top5 / n, "videos": self.n}


def evaluate(model, cfg, videos: Iterable[Tuple[torch.Tensor, int]], batch_videos: int = None) -> Dict[str, float]:
    """videos: iterable of (uint8 [F, H, W, 3] GPU tensor, label).  Batches `batch_videos` videos
    (default cfg.TEST.BATCH_SIZE) of views x crops clips each through `model(clips, training=False)`."""
    bv = int(batch_videos or cfg.TEST.BATCH_SIZE)
    nv = num_views(cfg)
    m = Metrics()
    clips, labels = [], []

    def flush():
        if clips:
            probs = model(torch.cat(clips, 0), training=False)
            m.update(probs.float(), torch.tensor(labels))
            clips.clear()
            labels.clear()

    for video, label in videos:
        c = make_eval_views(video, cfg, dtype=model.dtype)
        assert c.shape[0] == nv
        clips.append(c)
        labels.append(int(label))
        if len(clips) == bv:
            flush()
    flush()
    return m.result()
