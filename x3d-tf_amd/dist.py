"""Data parallelism for the X3D train step: one process per GPU, RCCL over xGMI.

The reference's only parallelism is tf.distribute.MirroredStrategy (reference utils.py:144-174): the
global batch is split across in-process replicas, BatchNorm statistics stay per replica, gradients
are all-reduced once per step and mirrored variables (BN moving statistics) are mean-aggregated.
Here: rank r owns clips [r*B, (r+1)*B) of the global batch; there is no data-path collective in the
forward or backward kernels; the flat gradient buffer is all-reduced in per-stage buckets, each
launched as soon as the backward pass has finished that stage (deepest stage first), so the ring
traffic (15 MB total for X3D-M; ~0.2 ms on a 153 GB/s xGMI link) hides under the remaining backward
depthwise/pointwise kernels.  ``backend="nccl"`` is RCCL on ROCm; the same code runs on gloo/CPU
tensors for the multi-process tests.
"""
import collections
import os
import time
from typing import List, Optional, Sequence, Tuple

# the host driver only supports dmabuf IPC; the HSA runtime reads this when HIP initialises, i.e. it has to be in the
# environment before the first torch.cuda call of the process (launchers export it too)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def visible_gpu_count(sysfs_root: str = "/sys/class/kfd/kfd/topology/nodes", dev_root: str = "/dev/dri") -> Optional[int]:
    """GPUs this process could open, counted WITHOUT loading the HIP / HSA runtime: the KFD topology in sysfs (nodes with
    simd_count > 0 whose render node exists and is accessible), then the *_VISIBLE_DEVICES filters.  None when the
    topology cannot be read (no amdgpu driver in this namespace) -- callers then leave the check to the ranks.

    For a launcher parent: torch.cuda.device_count() falls through to hipGetDeviceCount when amdsmi is not importable,
    which opens /dev/kfd, and a process that initialised the GPU must not fork-exec its ranks on this pool."""
    try:
        nodes = sorted(os.listdir(sysfs_root), key=lambda d: (len(d), d))
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            with open(os.path.join(sysfs_root, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue            # a node this cgroup may not read is a device it may not use
        if int(props.get("simd_count", "0")) <= 0:
            continue            # CPU node
        minor = props.get("drm_render_minor")
        if minor is not None and int(minor) > 0:
            if not os.access(os.path.join(dev_root, f"renderD{int(minor)}"), os.R_OK | os.W_OK):
                continue
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is None:
            continue
        ids = [t.strip() for t in v.split(",") if t.strip()]
        k = 0
        for t in ids:           # the runtime stops at the first invalid entry
            if t.startswith("GPU-") or (t.lstrip("-").isdigit() and 0 <= int(t) < n):
                k += 1
            else:
                break
        n = min(n, k)
    return n


def local_device(local_rank: int) -> int:
    """Device index of a local rank: one GPU per rank; ranks wrap around only in a gloo rehearsal."""
    n = torch.cuda.device_count()
    if n == 0:
        return 0
    if local_rank >= n and os.environ.get("X3D_DIST_BACKEND") != "gloo":
        raise RuntimeError(f"LOCAL_RANK {local_rank} but only {n} GPU(s) visible: one process per GPU")
    return local_rank % n


def _rehearse() -> bool:
    """X3D_DIST_REHEARSE=1: run every collective even with ONE rank (a one-GPU box then exercises the real RCCL
    communicator, the bucketed asynchronous all-reduces and their stream ordering; the sums are identities)."""
    return os.environ.get("X3D_DIST_REHEARSE") == "1"


def _active(group=None) -> bool:
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or _rehearse())


def init_process_group(backend: Optional[str] = None):
    """Initialise torch.distributed from MASTER_ADDR/MASTER_PORT/RANK/WORLD_SIZE if WORLD_SIZE > 1."""
    rank, local_rank, world = env_world()
    if (world > 1 or _rehearse()) and not dist.is_initialized():
        if backend is None:
            # X3D_DIST_BACKEND=gloo: rehearsal of the multi-process path on a box with fewer GPUs than ranks
            # (RCCL refuses two ranks on one device; gloo stages device tensors through the host)
            backend = os.environ.get("X3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_device(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Clips [lo, hi) of the global batch owned by `rank` (equal shards, as MirroredStrategy splits)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} replicas")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class BucketReducer:
    """Asynchronous sum all-reduce of a fixed list of buckets (contiguous slices of a flat buffer).

    ``launch(i)`` enqueues bucket i behind everything already on the current stream and returns at
    once (RCCL runs it on its own stream); ``finish()`` makes the current stream wait for all of them.
    With world_size 1 both are no-ops.
    """
    EVENT_WINDOW = 64          # timing-event pairs kept alive at most (exposed_ms)

    def __init__(self, buckets: Sequence[torch.Tensor], group=None):
        self.buckets = list(buckets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = _active(group)
        self._work: List = []
        self.launched = 0          # all-reduces enqueued so far (bench.py reports them)
        self.launched_bytes = 0
        self._t_mark, self._ev_mark = None, None
        # (start, done) event pairs of the most recent steps only: a training run that never asks for exposed_ms() must not
        # accumulate two HIP events per step without bound; completed pairs that fall out of the window are folded into sums
        self._pairs, self._host_s, self._host_n = collections.deque(), 0.0, 0
        self._dev_ms, self._dev_n = 0.0, 0

    def launch(self, i: int):
        if not self.active:
            return
        self.launched += 1
        self.launched_bytes += self.buckets[i].numel() * self.buckets[i].element_size()
        self._work.append(dist.all_reduce(self.buckets[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def mark_backward_done(self):
        """Call once the last backward launch of the step is on the current stream: records an event there (no
        synchronisation).  finish() measures from it how long the exchange stays EXPOSED, i.e. the part of the collectives
        that did not hide behind the backward pass."""
        if not self.active:
            return
        self._t_mark = time.perf_counter()
        self._ev_mark = None
        if self._device_events():
            self._ev_mark = torch.cuda.Event(enable_timing=True)
            self._ev_mark.record()

    def _device_events(self) -> bool:
        # RCCL: work.wait() makes the compute STREAM wait, so stream events bracket the exposed time without a host sync;
        # host-side backends (gloo) block the host in wait(): wall time then
        return bool(self.buckets) and self.buckets[0].is_cuda and dist.get_backend(self.group) == "nccl"

    def finish(self):
        for w in self._work:
            w.wait()
        self._work = []
        if not self.active or self._t_mark is None:
            return
        if self._ev_mark is not None:
            done = torch.cuda.Event(enable_timing=True)
            done.record()
            self._pairs.append((self._ev_mark, done))      # read later (exposed_ms), after the caller's own synchronisation
            while len(self._pairs) > self.EVENT_WINDOW:    # steps old: long finished, elapsed_time does not block
                a, b = self._pairs.popleft()
                if not b.query():                          # (not finished after all: keep it -- dropping it would bias the
                    self._pairs.appendleft((a, b))         #  average towards the steps that happened to be done)
                    break
                self._dev_ms += a.elapsed_time(b)
                self._dev_n += 1
        else:
            self._host_s += time.perf_counter() - self._t_mark
            self._host_n += 1
        self._t_mark, self._ev_mark = None, None

    def exposed_ms(self, reset: bool = True) -> Optional[float]:
        """Average over the steps since the last call of (every bucket landed) - (last backward kernel finished), in ms:
        HIP-event time on the compute stream with RCCL, host wall time of the waits with a host-side backend (there it
        also contains whatever backward work was still queued on the device).  ~0 when the exchange hid behind the backward
        pass; None when no multi-rank step was measured.  Call after a device synchronisation."""
        n = len(self._pairs) + self._dev_n + self._host_n
        if n == 0:
            return None
        total = sum(a.elapsed_time(b) for a, b in self._pairs) + self._dev_ms + 1e3 * self._host_s
        if reset:
            self._pairs.clear()
            self._host_s, self._host_n, self._dev_ms, self._dev_n = 0.0, 0, 0.0, 0
        return max(total / n, 0.0)


def broadcast_(tensors: Sequence[torch.Tensor], src: int = 0, group=None):
    """Make every replica start from rank `src`'s values (mirrored-variable initialisation)."""
    if _active(group):
        for t in tensors:
            dist.broadcast(t, src=src, group=group)


def mean_(t: torch.Tensor, group=None):
    """In-place mean over replicas (mirrored-variable MEAN aggregation of BN moving statistics) [TF-3p]."""
    if _active(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(dist.get_world_size(group))
    return t


def gather_over_ranks(value: float, device=None) -> List[float]:
    """`value` of every rank, in rank order (one entry without a process group)."""
    if not _active():
        return [value]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=device or ("cuda" if torch.cuda.is_available() else "cpu"))
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def max_over_ranks(value: float, device=None) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if torch.cuda.is_available() else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
