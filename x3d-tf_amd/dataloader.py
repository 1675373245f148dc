"""Host half of the input pipeline: the reference's ``dataloader.InputReader`` (reference
dataloader.py:11-197) and the record format of ``datasets/create_tfrecords.py`` (:48-141), without TensorFlow.

What the reference does                                             here
-----------------------------------------------------------------   ---------------------------------------------------
GZIP TFRecord files of tf.train.SequenceExample, one per video:     `read_tfrecords` (length | masked CRC32C | bytes |
  context video/num_frames, video/class/label (int64), feature        masked CRC32C, gzip stream), `parse_sequence_example`
  list `video` = one JPEG string per frame (create_tfrecords.py:      (protobuf wire format restated), `decode_jpeg` (PIL /
  48-82, dataloader.py:63-88: tf.image.decode_jpeg)                   libjpeg; tf.image.decode_jpeg's DCT method is [TF-3p])
text file of "<path> <label>" lines decoded with decord             `decode_video`: needs a `decoder` callable (decord is
  (dataloader.py:29-61), undecodable video -> zeros [100,240,144,3]    not installed here); the zeros replacement is kept
list_files(shuffle) -> interleave -> shuffle(16 * batch | 1024)     `InputReader.__call__`: the same stages as Python
  -> decode -> repeat (train) -> temporal + spatial transforms         generators, decode on a thread pool, the transforms on
  -> batch(drop_remainder) -> process_batch -> prefetch               the GPU (views.py: x3d_train_clip / x3d_eval_views),
  (dataloader.py:126-197)                                             batches prefetched by a background thread
`create_tfrecords.to_tf_example` / TFRecordWriter (GZIP, level 9)    `make_sequence_example`, `write_tfrecords`

The decoded video goes to the GPU as uint8 [F, H, W, 3]; everything after decoding is the device half (views.py).  The
order of a shuffled training stream is this module's own (seeded torch/NumPy generators): TF's is not reproducible [TF-3p].
"""
import glob
import gzip
import io
import queue
import struct
import threading
import warnings
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np
import torch

from .checkpoint import _get_varint, _parse_proto, _put_varint, crc32c, mask_crc

FAILED_VIDEO_SHAPE = (100, 240, 144, 3)   # dataloader.py:58-61: what an undecodable video is replaced with


# ------------------------------------------------------------------------------------------------
# TFRecord framing (tensorflow/core/lib/io/record_writer.cc) [TF-3p]:
#   uint64 length | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data)
# ------------------------------------------------------------------------------------------------
def read_tfrecords(path: str, compression: str = "GZIP", verify: bool = True) -> Iterator[bytes]:
    """Yield the raw records of one TFRecord file (dataloader.py:150-153: compression_type="GZIP")."""
    opener = gzip.open if compression == "GZIP" else open
    with opener(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError(f"{path}: truncated record header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify and mask_crc(crc32c(head[:8])) != lcrc:
                raise ValueError(f"{path}: corrupted record length")
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError(f"{path}: truncated record")
            if verify and mask_crc(crc32c(data)) != struct.unpack("<I", tail)[0]:
                raise ValueError(f"{path}: corrupted record data")
            yield data


def write_tfrecords(path: str, records: Iterable[bytes], compression: str = "GZIP", level: int = 9) -> int:
    """create_tfrecords.py:103-107: tf.io.TFRecordWriter(path, TFRecordOptions("GZIP", compression_level=9))."""
    opener = (lambda p: gzip.open(p, "wb", compresslevel=level)) if compression == "GZIP" else (lambda p: open(p, "wb"))
    n = 0
    with opener(path) as f:
        for rec in records:
            head = struct.pack("<Q", len(rec))
            f.write(head + struct.pack("<I", mask_crc(crc32c(head))) + rec + struct.pack("<I", mask_crc(crc32c(rec))))
            n += 1
    return n


# ------------------------------------------------------------------------------------------------
# tf.train.SequenceExample (tensorflow/core/example/{example,feature}.proto) [TF-3p]
#   SequenceExample { Features context = 1; FeatureLists feature_lists = 2; }
#   Features     { map<string, Feature> feature = 1; }          (map entry: key = 1, value = 2)
#   FeatureLists { map<string, FeatureList> feature_list = 1; }  FeatureList { repeated Feature feature = 1; }
#   Feature { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
#   BytesList { repeated bytes value = 1; }   Int64List { repeated int64 value = 1 [packed]; }
# ------------------------------------------------------------------------------------------------
def _len(field: int, payload: bytes) -> bytes:
    return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def _int64_feature(v: int) -> bytes:
    return _len(3, _len(1, _put_varint(v & 0xFFFFFFFFFFFFFFFF)))


def _bytes_feature(b: bytes) -> bytes:
    return _len(1, _len(1, b))


def _map_entry(key: str, value: bytes) -> bytes:
    return _len(1, _len(1, key.encode()) + _len(2, value))


def encode_jpeg(frame_u8: np.ndarray, quality: int = 90) -> bytes:
    """create_tfrecords.py:64-65: tf.image.encode_jpeg(frame, format='rgb', quality=90, optimize_size=True)."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(frame_u8), "RGB").save(buf, format="JPEG", quality=quality, optimize=True)
    return buf.getvalue()


def decode_jpeg(data: bytes) -> np.ndarray:
    """dataloader.py:85: tf.image.decode_jpeg -> uint8 [H, W, 3]."""
    from PIL import Image
    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def make_sequence_example(frames_u8: np.ndarray, class_id: int, quality: int = 90, encoded: Optional[List[bytes]] = None) -> bytes:
    """create_tfrecords.py:48-82 `to_tf_example`: frames [F, H, W, 3] uint8 + label -> serialized SequenceExample
    (`encoded`: already JPEG-encoded frames instead of `frames_u8`)."""
    jpegs = encoded if encoded is not None else [encode_jpeg(f, quality) for f in frames_u8]
    context = _map_entry("video/num_frames", _int64_feature(len(jpegs))) + _map_entry("video/class/label", _int64_feature(int(class_id)))
    flist = b"".join(_len(1, _bytes_feature(j)) for j in jpegs)
    return _len(1, context) + _len(2, _map_entry("video", flist))


def _int64_list(feature: bytes) -> List[int]:
    f = _parse_proto(feature)
    out: List[int] = []
    for lst in f.get(3, []):
        for v in _parse_proto(lst).get(1, []):
            if isinstance(v, (bytes, bytearray)):        # packed
                pos = 0
                while pos < len(v):
                    x, pos = _get_varint(v, pos)
                    out.append(x)
            else:
                out.append(v)
    return [x - (1 << 64) if x >= (1 << 63) else x for x in out]


def parse_sequence_example(buf: bytes) -> Tuple[List[bytes], int, int]:
    """dataloader.py:63-88 `parse_and_decode` up to the JPEG decode: -> (JPEG strings of the frames, num_frames, label);
    absent context features default to -1 as FixedLenFeature([], tf.int64, -1) does."""
    top = _parse_proto(buf)
    ctx: Dict[str, bytes] = {}
    for feats in top.get(1, []):
        for entry in _parse_proto(feats).get(1, []):
            e = _parse_proto(entry)
            ctx[e[1][0].decode()] = e[2][0]
    nf = _int64_list(ctx["video/num_frames"]) if "video/num_frames" in ctx else []
    lb = _int64_list(ctx["video/class/label"]) if "video/class/label" in ctx else []
    frames: List[bytes] = []
    for fl in top.get(2, []):
        for entry in _parse_proto(fl).get(1, []):
            e = _parse_proto(entry)
            if e[1][0].decode() != "video":
                continue
            for feat in _parse_proto(e[2][0]).get(1, []):
                for bl in _parse_proto(feat).get(1, []):
                    frames += _parse_proto(bl).get(1, [])
    return frames, (nf[0] if nf else -1), (lb[0] if lb else -1)


# ------------------------------------------------------------------------------------------------
class InputReader:
    """reference dataloader.py:11-27.  `reader(file_pattern, batch_size)` iterates over (clips, labels) batches on the GPU:
    training [B, T, S, S, 3] / [B]; evaluation [B * views * crops, T, S, S, 3] / [B] (dataloader.py:90-116).

    Additional arguments (the reference gets them from TF's globals): `device`, `dtype` (the compute type clips are cast
    to under mixed precision, dataloader.py:111-113), `seed`, `decoder` (path -> uint8 [F, H, W, 3] array for the
    non-TFRecord path), `num_workers` (decode threads), `prefetch` (batches kept ready).

    Data parallelism (`rank`, `world`; default: the torchrun environment).  The reference feeds ONE dataset of global
    batches to MirroredStrategy, which splits each batch over the replicas (train.py:145-152, utils.py:160-167).  Here
    every process owns its reader, so the split is made at the source: `batch_size` stays the reference's GLOBAL batch
    (cfg.TRAIN.BATCH_SIZE), rank r keeps records r, r + world, ... of the interleaved record stream (disjoint shards that
    cover the dataset) and yields batch_size // world clips per step -- `Trainer.fit`'s DATASET_SIZE // BATCH_SIZE steps
    are then one pass over the data, and the lr schedule sees the reference's epochs.  The file order is drawn from a
    generator seeded by `seed` alone (identical on every rank, or the shards would overlap: with world > 1 an unset seed
    becomes 0); shuffle buffer and augmentation draws use (seed, rank)."""

    def __init__(self, cfg, is_training: bool, use_tfrecord: bool, mixed_precision: bool = False, device=None,
                 dtype: torch.dtype = torch.float32, seed: Optional[int] = None, decoder: Optional[Callable] = None,
                 num_workers: int = 4, prefetch: int = 2, rank: Optional[int] = None, world: Optional[int] = None):
        self._cfg = cfg
        self._is_training = bool(is_training)
        self._use_tfrecord = bool(use_tfrecord)
        self._mixed_prec = bool(mixed_precision)
        self._device = torch.device(device if device is not None else "cuda")
        self._dtype = dtype if mixed_precision else torch.float32
        if rank is None or world is None:
            from .dist import env_world
            env_rank, _, env_size = env_world()
            rank = env_rank if rank is None else rank
            world = env_size if world is None else world
        self._rank, self._world = int(rank), max(1, int(world))
        if not 0 <= self._rank < self._world:
            raise ValueError(f"rank {rank} outside [0, {world})")
        if self._world > 1 and seed is None:
            seed = 0
        self._rng_files = np.random.default_rng(seed)                       # the same stream on every rank
        self._rng = np.random.default_rng(seed if self._world == 1 else [int(seed), self._rank])
        self._gen = torch.Generator()
        self._gen.manual_seed(int(self._rng.integers(0, 2 ** 31)))
        self._decoder = decoder
        self._workers = max(1, int(num_workers))
        self._prefetch = max(1, int(prefetch))
        self.last_params: List[dict] = []      # the random draws of the clips of the last training batch (tests)

    # -- decode ------------------------------------------------------------------------------------
    def decode_video(self, line: str) -> Tuple[np.ndarray, int]:
        """dataloader.py:29-61: "<path> <label>" -> (all frames uint8 [F, H, W, 3], label); an undecodable video becomes
        zeros [100, 240, 144, 3] with a warning, as in the reference."""
        parts = line.strip().split(" ")
        path, label = parts[0], int(parts[1])
        try:
            if self._decoder is None:
                raise RuntimeError("no video decoder configured (the reference uses decord, which is not installed): "
                                   "pass InputReader(..., decoder=callable) or use TFRecords")
            video = np.ascontiguousarray(self._decoder(path), dtype=np.uint8)
            if video.ndim != 4 or video.shape[-1] != 3:
                raise ValueError(f"decoder returned shape {video.shape}")
        except Exception as e:   # noqa: BLE001 -- the reference catches everything here
            warnings.warn(f"Failed to decode video {path} ({e}). Replacing with zeros...")
            video = np.zeros(FAILED_VIDEO_SHAPE, np.uint8)
        return video, label

    def parse_and_decode(self, serialized_example: bytes) -> Tuple[np.ndarray, int]:
        """dataloader.py:63-88: SequenceExample -> (video uint8 [num_frames, H, W, 3], label)."""
        jpegs, num_frames, label = parse_sequence_example(serialized_example)
        n = num_frames if num_frames >= 0 else len(jpegs)
        return np.stack([decode_jpeg(j) for j in jpegs[:n]]), int(label)

    # -- stages ------------------------------------------------------------------------------------
    def _shuffle(self, it: Iterator, size: int) -> Iterator:
        """tf.data shuffle: a buffer of `size` elements, a uniformly drawn one leaves when the next arrives."""
        buf: List = []
        for x in it:
            if len(buf) < size:
                buf.append(x)
                continue
            i = int(self._rng.integers(0, size))
            yield buf[i]
            buf[i] = x
        while buf:
            i = int(self._rng.integers(0, len(buf)))
            yield buf.pop(i)

    def _records(self, file_pattern: str, batch_size: Optional[int]) -> Iterator:
        files = sorted(glob.glob(file_pattern))
        if not files:
            raise FileNotFoundError(f"no files match {file_pattern}")
        if self._use_tfrecord:
            if self._is_training:
                files = [files[i] for i in self._rng_files.permutation(len(files))]     # list_files(shuffle=True)
            its = [read_tfrecords(f) for f in files]

            def interleave():                    # cycle over the open files, one record each (dataloader.py:149-155)
                live = list(its)
                while live:
                    for it in list(live):
                        try:
                            yield next(it)
                        except StopIteration:
                            live.remove(it)
            recs = self._shard(interleave(), batch_size)
            if self._is_training:
                recs = self._shuffle(recs, batch_size * 16 if batch_size else 1024)
            return recs
        lines = [ln for f in files for ln in open(f).read().splitlines() if ln.strip()]   # TextLineDataset(...).cache()
        if self._is_training:
            # .shuffle(DATASET_SIZE, reshuffle_each_iteration): a fresh permutation of the WHOLE list every pass, drawn from
            # the rank-independent generator BEFORE the split -- every rank sees every video over the epochs, as the
            # replicas of the reference's one shuffled dataset do (a per-rank shuffle of a fixed shard would not)
            lines = [lines[i] for i in self._rng_files.permutation(len(lines))]
        return self._shard(iter(lines), batch_size)

    def _shard(self, it: Iterator, local_batch: Optional[int] = None) -> Iterator:
        """this rank's records of the (rank-independent) record stream, cut so that every rank gets the SAME number:
        training -- records r, r + world, ... of the complete groups of `world` records (tf.data's `shard`; the
        pass's last N % world records are dropped, so all ranks cross pass boundaries at the same step and keep drawing the
        same per-pass permutations); evaluation -- the stream is cut into GLOBAL batches of world * local_batch records,
        rank r takes records [r * b, (r + 1) * b) of each complete one and the trailing partial global batch is dropped:
        exactly the videos `batch(drop_remainder=True)` of the reference's single dataset evaluates."""
        if self._world == 1:
            return it
        group = self._world if self._is_training or not local_batch else self._world * local_batch
        per = group // self._world

        def gen():
            buf = []
            for x in it:
                buf.append(x)
                if len(buf) == group:
                    if self._is_training:
                        yield buf[self._rank]
                    else:
                        yield from buf[self._rank * per:(self._rank + 1) * per]
                    buf = []
        return gen()

    def local_batch(self, batch_size: Optional[int]) -> Optional[int]:
        """clips (videos in evaluation) per step on this rank for the reference's global `batch_size`"""
        if batch_size is None or self._world == 1:
            return batch_size
        if batch_size % self._world:
            raise ValueError(f"global batch {batch_size} is not divisible by {self._world} replicas")
        return batch_size // self._world

    def _decoded(self, recs_factory: Callable[[], Iterator]) -> Iterator[Tuple[np.ndarray, int]]:
        fn = self.parse_and_decode if self._use_tfrecord else self.decode_video
        with ThreadPoolExecutor(self._workers) as ex:
            while True:
                pending: "queue.Queue" = queue.Queue()
                n = 0
                for rec in recs_factory():
                    pending.put(ex.submit(fn, rec))
                    n += 1
                    if pending.qsize() >= 2 * self._workers:
                        yield pending.get().result()
                while not pending.empty():
                    yield pending.get().result()
                if not self._is_training or n == 0:     # dataset.repeat() only in training (dataloader.py:171-172)
                    return

    def _clips(self, video: np.ndarray) -> torch.Tensor:
        from .views import draw_train_params, make_eval_views, make_train_clip
        v = torch.from_numpy(video).to(self._device, non_blocking=True)
        if self._is_training:
            f, h, w, _ = video.shape
            p = draw_train_params(f, h, w, self._cfg, self._gen)
            self._batch_params.append(p)
            return make_train_clip(v, self._cfg, params=p, dtype=self._dtype)[None]
        return make_eval_views(v, self._cfg, dtype=self._dtype)

    def process_batch(self, clips: List[torch.Tensor], labels: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        """dataloader.py:90-116: [B, T, S, S, 3] in training, [B * views * crops, T, S, S, 3] otherwise."""
        return torch.cat(clips, 0), torch.tensor(labels, dtype=torch.int64, device=self._device)

    def _batches(self, file_pattern: str, batch_size: Optional[int]) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        clips: List[torch.Tensor] = []
        labels: List[int] = []
        self._batch_params: List[dict] = []
        bs = batch_size or 1
        for video, label in self._decoded(lambda: self._records(file_pattern, batch_size)):
            clips.append(self._clips(video))
            labels.append(label)
            if len(clips) == bs:
                params, self._batch_params = self._batch_params, []
                yield self.process_batch(clips, labels) + (params,)
                clips, labels = [], []
        # drop_remainder=True (dataloader.py:186): a trailing partial batch is not emitted

    def __call__(self, file_pattern: str, batch_size: Optional[int] = None) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """dataloader.py:126-197.  Iterate to get batches; a background thread keeps `prefetch` of them ready.
        `batch_size` is the GLOBAL batch; with world > 1 this rank yields its batch_size // world share (class docstring)."""
        batch_size = self.local_batch(batch_size)
        q: "queue.Queue" = queue.Queue(self._prefetch)
        stop = threading.Event()
        END = object()

        def work():
            try:
                if self._device.type == "cuda":
                    torch.cuda.set_device(self._device)
                for b in self._batches(file_pattern, batch_size):
                    if self._device.type == "cuda":
                        torch.cuda.current_stream().synchronize()     # the consumer thread uses another stream context
                    while not stop.is_set():
                        try:
                            q.put(b, timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(END)
            except BaseException as e:   # noqa: BLE001 -- re-raised in the consumer
                q.put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    return
                if isinstance(item, BaseException):
                    raise item
                self.last_params = item[2]      # the draws of THIS batch (the producer thread runs `prefetch` batches ahead)
                yield item[0], item[1]
        finally:
            stop.set()
