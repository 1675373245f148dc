"""Evaluation driver: the part of the reference's eval.py (:75-89) that runs after the model is built --
`model.load_weights(latest_checkpoint).expect_partial()`, then `model.evaluate(...)` with
SparseCategoricalCrossentropy, SparseCategoricalAccuracy ('acc') and SparseTopKCategoricalAccuracy(k=5)
('top_5_acc') (eval.py:48-66).  Videos come as decoded uint8 tensors; the views are built on the GPU (views.py)."""
from typing import Dict, Iterable, Tuple

import torch

from .views import make_eval_views, num_views


class Metrics:
    """Running means in the Keras sense: per-video loss / hits averaged over the videos seen so far."""

    def __init__(self, reg_loss: float = 0.0):
        self.reg_loss = float(reg_loss)   # the model's L2 term: Keras adds it to every reported `loss`
        self.n = 0
        self.loss = 0.0
        self.top1 = 0
        self.top5 = 0

    def update(self, probs: torch.Tensor, labels: torch.Tensor):
        """probs [videos, classes] fp32 (already view-averaged by the model); labels [videos]."""
        labels = labels.to(probs.device).long()
        # Keras CE from probabilities (the same expression as x3d_softmax_xent and the oracle):
        # q = clip(p, 1e-7, 1 - 1e-7); loss = -log q_y + log sum_j q_j
        q = probs.double().clamp(1e-7, 1.0 - 1e-7)
        self.loss += float((-q.gather(1, labels[:, None]).squeeze(1).log() + q.sum(1).log()).sum())
        top5 = probs.topk(min(5, probs.shape[1]), dim=1).indices
        self.top1 += int((top5[:, 0] == labels).sum())
        self.top5 += int((top5 == labels[:, None]).any(dim=1).sum())
        self.n += int(labels.numel())

    def all_reduce_(self, group=None, device=None):
        """Sum the numerators and the video count over the replicas (no-op without a multi-rank process group): under
        MirroredStrategy `model.evaluate` reports ONE metric over the whole dataset (reference eval.py:83-89,
        utils.py:160-167); here every rank evaluates its shard of the videos and only these four counters are exchanged
        ("eval -- replicas only ... only metric counters are reduced")."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        if device is None:
            device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([self.loss, float(self.top1), float(self.top5), float(self.n)], dtype=torch.float64, device=device)
        dist.all_reduce(t, group=group)
        self.loss, self.top1, self.top5, self.n = float(t[0]), int(round(float(t[1]))), int(round(float(t[2]))), int(round(float(t[3])))
        return self

    def result(self) -> Dict[str, float]:
        n = max(self.n, 1)
        return {"loss": self.loss / n + self.reg_loss, "acc": self.top1 / n, "top_5_acc": self.top5 / n, "videos": self.n}


def evaluate(model, cfg, videos: Iterable[Tuple[torch.Tensor, int]], batch_videos: int = None) -> Dict[str, float]:
    """videos: iterable of (uint8 [F, H, W, 3] GPU tensor, label).  Batches `batch_videos` videos
    (default cfg.TEST.BATCH_SIZE) of views x crops clips each through `model(clips, training=False)`."""
    bv = int(batch_videos or cfg.TEST.BATCH_SIZE)
    nv = num_views(cfg)
    # `model.evaluate` reports cross-entropy + the model's regularisation losses (weight_decay * sum w^2, model.py:47)
    m = Metrics(float(model.regularization_loss().item()) if hasattr(model, "regularization_loss") else 0.0)
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
    if hasattr(model, "release_plans"):
        model.release_plans(keep=1)     # the partial tail batch allocated a second multi-GB plan
    return m.result()


def evaluate_dataset(model, cfg, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], group=None) -> Dict[str, float]:
    """reference eval.py:83-89 `model.evaluate(InputReader(cfg, False, use_tfrecord)(pattern, cfg.TEST.BATCH_SIZE))`:
    `batches` yields (clips [B * views * crops, T, S, S, 3], labels [B]) as `dataloader.InputReader` does in evaluation
    mode (the views were built on the GPU while the batch was assembled).  Under torchrun the reader hands every rank its
    share of each global batch; the counters are summed over the ranks ONCE at the end, so every rank returns the metric of
    the whole dataset (the same videos the single-process run evaluates: the reader drops the same trailing partial batch)."""
    m = Metrics(float(model.regularization_loss().item()) if hasattr(model, "regularization_loss") else 0.0)
    dev = None
    for clips, labels in batches:
        m.update(model(clips, training=False).float(), labels)
        dev = clips.device
    if hasattr(model, "release_plans"):
        model.release_plans(keep=1)
    m.all_reduce_(group, dev)
    return m.result()
