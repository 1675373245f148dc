"""The train step the reference compiles with Keras (reference train.py:85-125): SGD with Nesterov
momentum, SparseCategoricalCrossentropy on the model's probabilities + L2 regularisation, per-epoch
warm-up/cosine learning-rate schedule -- as an explicit step loop over the HIP model, data-parallel
over RCCL when launched with torchrun.
"""
import math
import os
from typing import Optional

import torch

from . import dist as xdist


def lr_schedule(epoch, cfg):
    """reference train.py:114-125: linear warm-up while epoch <= WARMUP_EPOCHS, then half-cosine."""
    tr = cfg.TRAIN
    if epoch > tr.WARMUP_EPOCHS:
        return tr.BASE_LR * (0.5 * (math.cos(math.pi * (epoch / tr.EPOCHS)) + 1))
    return tr.WARMUP_LR + epoch * (tr.BASE_LR - tr.WARMUP_LR) / tr.WARMUP_EPOCHS


class Trainer:
    """fwd + bwd + gradient all-reduce + optimizer for one replica.

    model: x3d_tf_amd.model.X3D.  The process group (if any) must already be initialised; every
    replica is given rank 0's initial variables, as variables created under MirroredStrategy are.
    """

    def __init__(self, model, cfg, momentum: Optional[float] = None, sync_moving_stats: bool = True, group=None,
                 loss_scale="auto"):
        self.optimizer = cfg.TRAIN.OPTIMIZER.lower()
        if self.optimizer not in ("sgd", "adam"):   # reference train.py:88-97: SGD(nesterov) / Adam / NotImplementedError
            raise NotImplementedError(f"{cfg.TRAIN.OPTIMIZER} not supported")
        self.model, self.cfg, self.group = model, cfg, group
        self.opt_step = 0                     # optimizer steps applied (Adam's bias correction counts them)
        # Loss scaling = tf.keras.mixed_precision.LossScaleOptimizer(opt) with its defaults (train.py:99-100): dynamic,
        # initial scale 2^15, doubled after 2000 consecutive finite steps, halved (and the step skipped) when a gradient
        # is inf / nan.  "auto": on for float16 storage (the reference's mixed_float16), off for float32 / bfloat16;
        # a number = fixed scale; None = off.
        if loss_scale == "auto":
            loss_scale = "dynamic" if model.dtype == torch.float16 else None
        self.dynamic_scale = loss_scale == "dynamic"
        self.loss_scale = 2.0 ** 15 if self.dynamic_scale else (float(loss_scale) if loss_scale else 1.0)
        self.growth_steps, self._good_steps, self.skipped_steps = 2000, 0, 0
        self.momentum = cfg.TRAIN.MOMENTUM if momentum is None else momentum
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
        self.collectives = xdist._active(group)      # world > 1, or the one-rank rehearsal (X3D_DIST_REHEARSE=1)
        self.sync_moving_stats = sync_moving_stats and self.collectives
        n_st = len(model.arch.stages)
        # bucket order = order in which the backward pass finishes them: head, stage 3..0, stem.  (Merging them into two
        # all-reduces -- {head + last stage}, {rest} -- was measured with X3D_DIST_REHEARSE=1: 26.19 -> 26.16 ms per
        # step, i.e. the per-call cost is not what the 0.45 ms of collective overhead on one rank is made of.)
        per_stage = [n_st] + list(range(n_st - 1, -1, -1)) + [-1]
        self.stage_order = per_stage
        buckets = [model.grad_bucket(s) for s in per_stage]
        self._launch_at = {s: i for i, s in enumerate(per_stage)}
        self.reducer = xdist.BucketReducer(buckets, group)
        self._slot = dict(self._launch_at)
        self._stats_work = None
        xdist.broadcast_([model.flat_params, model.flat_velocity], 0, group)
        self.epoch = 0

    def step(self, clips, labels, lr: Optional[float] = None):
        """clips: this replica's shard [B, T, H, W, 3]; labels [B].  Returns the plan (loss_rows, probs)."""
        m = self.model
        n = clips.shape[0]
        if lr is None:
            lr = lr_schedule(self.epoch, self.cfg)
        pl = m.forward_backward(clips, labels, global_batch=n * self.world,
                                on_stage_done=self._on_stage_done if self.collectives else None,
                                loss_scale=self.loss_scale)
        self.reducer.mark_backward_done()     # (an event on the compute stream: finish() measures the exposed exchange from it)
        self.reducer.finish()
        if self._stats_work is not None:      # mirrored-variable MEAN aggregation of the BN moving statistics [TF-3p]
            self._stats_work.wait()
            self._stats_work = None
            m.moving_stats_flat().div_(self.world)
        if self.dynamic_scale:                # after the all-reduce: every replica sees the same sums, takes the same branch
            if not m.grads_finite():
                self.loss_scale = max(self.loss_scale / 2.0, 1.0)
                self._good_steps = 0
                self.skipped_steps += 1
                return pl                     # LossScaleOptimizer skips the update
            self._good_steps += 1
        self.opt_step += 1
        if self.optimizer == "adam":
            m.apply_adam(lr, self.opt_step, grad_scale=1.0 / self.loss_scale)
        else:
            m.apply_sgd(lr, self.momentum, grad_scale=1.0 / self.loss_scale)
        if self.dynamic_scale and self._good_steps >= self.growth_steps:
            self.loss_scale *= 2.0
            self._good_steps = 0
        return pl

    def fit(self, dataset, epochs: Optional[int] = None, steps_per_epoch: Optional[int] = None, model_dir: Optional[str] = None,
            initial_epoch: Optional[int] = None, on_step=None):
        """The loop `model.fit(dataset, epochs, steps_per_epoch, initial_epoch, callbacks)` runs in reference train.py:145-152
        with its LearningRateScheduler (per-epoch `lr_schedule`, train.py:114-125) and ModelCheckpoint (`ckpt-{epoch}`
        after every epoch, utils.py:128-132) callbacks -- nothing else of the Keras harness (TensorBoard / wandb callbacks
        are out of scope).  `dataset`: an iterator of (clips, labels) batches, e.g. `dataloader.InputReader(cfg, True,
        True)(pattern, cfg.TRAIN.BATCH_SIZE)` (infinite in training mode, like `dataset.repeat()`; under torchrun the reader
        shards the records by rank and yields BATCH_SIZE // world clips per step, so `steps_per_epoch` = DATASET_SIZE //
        BATCH_SIZE is one pass over the data on every world size, as with MirroredStrategy).  Returns the per-epoch mean
        losses."""
        tr = self.cfg.TRAIN
        epochs = int(tr.EPOCHS if epochs is None else epochs)
        steps = int(steps_per_epoch if steps_per_epoch is not None else tr.DATASET_SIZE // tr.BATCH_SIZE)
        if steps <= 0:
            raise ValueError("steps_per_epoch must be positive (cfg.TRAIN.DATASET_SIZE // cfg.TRAIN.BATCH_SIZE)")
        if initial_epoch is not None:
            self.epoch = int(initial_epoch)
        it = iter(dataset)
        history = []
        while self.epoch < epochs:
            lr = lr_schedule(self.epoch, self.cfg)
            tot = torch.zeros((), dtype=torch.float64, device=self.model.device)
            for _ in range(steps):
                clips, labels = next(it)
                pl = self.step(clips, labels, lr)
                tot += self.loss(pl).double()
                if on_step is not None:
                    on_step(self, pl)
            self.epoch += 1
            history.append(float(tot.item()) / steps)
            if model_dir is not None and xdist.env_world()[0] == 0:
                self.save_checkpoint(model_dir, self.epoch)
        return history

    def collective_stats(self):
        """What the exchange step of this replica did so far (bench.py's `collectives` block)."""
        r = self.reducer
        return {"backend": (torch.distributed.get_backend(self.group) if torch.distributed.is_initialized() else None),
                "world": self.world, "buckets_per_step": len(r.buckets),
                "bytes_per_step": sum(b.numel() * b.element_size() for b in r.buckets),
                "allreduces_launched": r.launched, "allreduce_bytes": r.launched_bytes,
                "launched_from_backward_hooks": bool(self.collectives and r.launched > 0),
                "moving_stats_mean": bool(self.sync_moving_stats),
                # time between the end of the backward pass and the last bucket landing, averaged over the steps since the
                # last call (0 = fully hidden; None = single rank); "device" = HIP events on the compute stream (RCCL)
                "exposed_ms": r.exposed_ms(),
                "exposed_clock": ("device" if (r.active and r._device_events()) else ("host" if r.active else None))}

    def _on_stage_done(self, stage):
        """backward hook (model.forward_backward): 'fwd' = forward finished (the moving statistics are final: their
        all-reduce overlaps the whole backward pass), else a stage whose gradients are final."""
        if stage == "fwd":
            if self.sync_moving_stats:
                self._stats_work = torch.distributed.all_reduce(self.model.moving_stats_flat(), group=self.group,
                                                                async_op=True)
            return
        i = self._launch_at.get(stage)
        if i is not None:
            self.reducer.launch(i)

    # -- checkpoints in the reference's layout (utils.py:128-132 ModelCheckpoint 'ckpt-{epoch:d}', train.py:131-136) --
    def save_checkpoint(self, model_dir: str, epoch: int) -> str:
        """Writes `<model_dir>/ckpt-<epoch>` as a TF tensor bundle -- weights, the optimizer's slot variables (SGD:
        `momentum`; Adam: `m` and `v`), its hyper-parameter variables (`iter`, `learning_rate`, `decay` + `momentum` |
        `beta_1`, `beta_2`) and the `_CHECKPOINTABLE_OBJECT_GRAPH` Keras' `load_weights` restores by -- and the
        `checkpoint` state file, so that the reference's `tf.train.latest_checkpoint(model_dir)` finds it (ModelCheckpoint
        saves the whole optimizer, utils.py:128-132)."""
        import os
        os.makedirs(model_dir, exist_ok=True)
        prefix = os.path.join(model_dir, f"ckpt-{int(epoch)}")
        hyper = dict(iter=self.opt_step, learning_rate=lr_schedule(self.epoch, self.cfg), decay=0.0)
        if self.optimizer == "adam":
            hyper.update(beta_1=0.9, beta_2=0.999)
        else:
            hyper.update(momentum=self.momentum)
        self.model.save_weights(prefix, optimizer_hyper=hyper, optimizer=self.optimizer)
        return prefix

    def resume(self, model_dir: str) -> int:
        """train.py:131-136: load the latest `ckpt-<epoch>` of `model_dir` if there is one; returns the epoch to
        continue from (0 without a checkpoint) and sets `self.epoch`."""
        import os
        from .checkpoint import latest_checkpoint
        path = latest_checkpoint(model_dir)
        if not path:
            return 0
        m = self.model
        m.load_weights(path, optimizer=self.optimizer)   # weights + this branch's optimizer slots; unknown keys tolerated as Keras does
        st = getattr(m, "optimizer_state", None) or {}
        kind = st.get("kind")
        if kind is not None and kind != self.optimizer:
            # a checkpoint written by the other optimizer branch: Keras restores the variables and leaves the new
            # optimizer's slots at their initial value -- never reuse SGD momentum as Adam's first moment or vice versa
            m.flat_velocity.zero_()
            if getattr(m, "flat_second", None) is not None:
                m.flat_second.zero_()
            self.opt_step = 0
        else:
            # optimizer/iter: Adam's bias correction continues from the saved step count
            self.opt_step = int(st.get("hyper", {}).get("iter", 0))
            if self.optimizer == "adam" and kind is None:
                m.flat_velocity.zero_()
                self.opt_step = 0
        self.epoch = int(os.path.basename(path).split("-")[1])
        return self.epoch

    def loss(self, pl):
        """global-batch mean cross-entropy + L2 term (what Keras reports as `loss`)."""
        ce = pl.loss_rows.sum() / (pl.n * self.world)
        if self.collectives:
            torch.distributed.all_reduce(ce, group=self.group)
        return ce + self.model.regularization_loss().float().squeeze()
