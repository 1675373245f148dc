"""Host half of the input pipeline (x3d_tf_amd/dataloader.py = reference dataloader.py + datasets/create_tfrecords.py):
record framing, the SequenceExample wire format against Google's protobuf runtime as an independent encoder / decoder,
JPEG decode, shuffle / batch semantics on the CPU; the whole pipeline against oracle/views_oracle.py on the GPU."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from x3d_tf_amd import dataloader as DL  # noqa: E402


def _videos(n, seed=0, f_lo=5, f_hi=9, h=24, w=32):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        f = int(rng.integers(f_lo, f_hi))
        # smooth content (JPEG-friendly) + a per-video offset so that videos are distinguishable after lossy coding
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.sin(yy / 5.0 + i)[..., None] * 60 + np.cos(xx[..., None] / 7.0 + np.arange(3) + i) * 60 + 128   # [h, w, 3]
        vid = np.stack([np.clip(base + 10 * t, 0, 255) for t in range(f)]).astype(np.uint8)
        out.append((vid, int(rng.integers(0, 400))))
    return out


def _example_proto_classes():
    """tf.train.SequenceExample & co. rebuilt from the published .proto (tensorflow/core/example/{example,feature}.proto)
    with Google's protobuf runtime: an implementation of the wire format that shares no code with dataloader.py."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="x3d_test_example.proto", package="x3dtest", syntax="proto3")

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=F.LABEL_OPTIONAL, type_name=None, oneof=None):
        f = m.field.add(name=name, number=num, type=typ, label=label)
        if type_name:
            f.type_name = ".x3dtest." + type_name
        if oneof is not None:
            f.oneof_index = oneof
        return f

    field(msg("BytesList"), "value", 1, F.TYPE_BYTES, F.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, F.TYPE_FLOAT, F.LABEL_REPEATED)
    field(msg("Int64List"), "value", 1, F.TYPE_INT64, F.LABEL_REPEATED)
    feat = msg("Feature")
    feat.oneof_decl.add(name="kind")
    field(feat, "bytes_list", 1, F.TYPE_MESSAGE, type_name="BytesList", oneof=0)
    field(feat, "float_list", 2, F.TYPE_MESSAGE, type_name="FloatList", oneof=0)
    field(feat, "int64_list", 3, F.TYPE_MESSAGE, type_name="Int64List", oneof=0)
    feats = msg("Features")
    e = feats.nested_type.add(name="FeatureEntry")
    e.options.map_entry = True
    e.field.add(name="key", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    e.field.add(name="value", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name=".x3dtest.Feature")
    feats.field.add(name="feature", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED,
                    type_name=".x3dtest.Features.FeatureEntry")
    field(msg("FeatureList"), "feature", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, type_name="Feature")
    fls = msg("FeatureLists")
    e = fls.nested_type.add(name="FeatureListEntry")
    e.options.map_entry = True
    e.field.add(name="key", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    e.field.add(name="value", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name=".x3dtest.FeatureList")
    fls.field.add(name="feature_list", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED,
                  type_name=".x3dtest.FeatureLists.FeatureListEntry")
    se = msg("SequenceExample")
    field(se, "context", 1, F.TYPE_MESSAGE, type_name="Features")
    field(se, "feature_lists", 2, F.TYPE_MESSAGE, type_name="FeatureLists")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("x3dtest.SequenceExample"))


def test_tfrecord_framing_roundtrip_and_corruption(tmp_path):
    recs = [b"", b"x", os.urandom(1000), b"abc" * 50]
    for comp in ("GZIP", ""):
        p = str(tmp_path / f"a{comp}.tfrecord")
        assert DL.write_tfrecords(p, recs, compression=comp) == 4
        assert list(DL.read_tfrecords(p, compression=comp)) == recs
    raw = bytearray(open(str(tmp_path / "a.tfrecord"), "rb").read())
    # known layout of the first (empty) record: length 0, masked crc of eight zero bytes, masked crc of b""
    assert raw[:8] == b"\0" * 8 and len(raw) == sum(16 + len(r) for r in recs)
    raw[16 + 12] ^= 0xFF          # first data byte of record 1
    open(str(tmp_path / "bad.tfrecord"), "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="corrupted record data"):
        list(DL.read_tfrecords(str(tmp_path / "bad.tfrecord"), compression=""))
    open(str(tmp_path / "short.tfrecord"), "wb").write(bytes(raw[:-3]))
    with pytest.raises(ValueError, match="truncated"):
        list(DL.read_tfrecords(str(tmp_path / "short.tfrecord"), compression="", verify=False))


def test_sequence_example_against_protobuf_runtime():
    SE = _example_proto_classes()
    jpegs = [b"\xff\xd8frame0", b"\xff\xd8frame-one", b""]
    # (1) what this module writes is what protobuf reads
    m = SE()
    m.ParseFromString(DL.make_sequence_example(None, 321, encoded=jpegs))
    assert list(m.context.feature["video/num_frames"].int64_list.value) == [3]
    assert list(m.context.feature["video/class/label"].int64_list.value) == [321]
    assert [f.bytes_list.value[0] for f in m.feature_lists.feature_list["video"].feature] == jpegs
    # (2) what protobuf writes (packed int64, its own field order, an extra feature) is what this module reads
    m2 = SE()
    m2.context.feature["video/class/label"].int64_list.value.append(7)
    m2.context.feature["video/num_frames"].int64_list.value.append(2)
    m2.context.feature["something/else"].bytes_list.value.append(b"ignored")
    for j in jpegs[:2]:
        m2.feature_lists.feature_list["video"].feature.add().bytes_list.value.append(j)
    m2.feature_lists.feature_list["audio"].feature.add().bytes_list.value.append(b"not video")
    frames, nf, label = DL.parse_sequence_example(m2.SerializeToString())
    assert frames == jpegs[:2] and nf == 2 and label == 7
    # absent context features default to -1 (FixedLenFeature([], tf.int64, -1)); negative labels survive the varint
    assert DL.parse_sequence_example(b"") == ([], -1, -1)
    assert DL.parse_sequence_example(DL.make_sequence_example(None, -5, encoded=[]))[2] == -5


def test_jpeg_and_parse_and_decode():
    (vid, label), = _videos(1, seed=3)
    ex = DL.make_sequence_example(vid, label, quality=90)
    rd = DL.InputReader.__new__(DL.InputReader)
    video, lab = DL.InputReader.parse_and_decode(rd, ex)
    assert video.dtype == np.uint8 and video.shape == vid.shape and lab == label
    assert np.abs(video.astype(int) - vid.astype(int)).mean() < 3.0          # JPEG quality 90 on smooth content
    jpegs, nf, _ = DL.parse_sequence_example(ex)
    assert nf == len(vid) and np.array_equal(video[2], DL.decode_jpeg(jpegs[2]))


def test_shuffle_buffer_and_failed_decode():
    import x3d_tf_amd as x
    cfg = x.get_config("XS")
    rd = DL.InputReader(cfg, True, True, device="cpu", seed=5)
    out = list(rd._shuffle(iter(range(100)), 16))
    assert sorted(out) == list(range(100)) and out != list(range(100))
    assert all(v <= i + 16 for i, v in enumerate(out))     # output i is drawn from the first i + 16 elements read so far
    assert list(rd._shuffle(iter(range(10)), 1)) == list(range(10))
    # undecodable video -> zeros [100, 240, 144, 3] + warning (dataloader.py:55-61); a decoder is honoured
    with pytest.warns(UserWarning, match="Failed to decode"):
        v, lab = rd.decode_video("/no/such/file.mp4 17\n")
    assert v.shape == DL.FAILED_VIDEO_SHAPE and v.dtype == np.uint8 and not v.any() and lab == 17
    rd2 = DL.InputReader(cfg, False, False, device="cpu", decoder=lambda p: np.full((3, 8, 8, 3), 9, np.uint8))
    v, lab = rd2.decode_video("clip.mp4 2")
    assert v.shape == (3, 8, 8, 3) and int(v[0, 0, 0, 0]) == 9 and lab == 2


def test_rank_shards_are_disjoint_and_cover_the_dataset(tmp_path):
    """MirroredStrategy splits ONE global batch over the replicas (reference utils.py:160-167); here each process owns a
    reader, so rank r keeps records r, r + world, ... of the same record stream: disjoint, covering, batch_size // world
    per step -- in evaluation order and through the training shuffles (file order is rank-independent)."""
    import x3d_tf_amd as x
    cfg = x.get_config("XS")
    recs = [bytes([i]) * (3 + i) for i in range(14)]
    for k in range(0, 14, 4):
        DL.write_tfrecords(str(tmp_path / f"part-{k // 4}.tfrecord"), recs[k:k + 4])
    pattern = str(tmp_path / "part-*.tfrecord")
    one = list(DL.InputReader(cfg, False, True, device="cpu", rank=0, world=1)._records(pattern, 8))
    assert sorted(one) == sorted(recs)
    for training in (False, True):
        shards = []
        for r in range(2):
            rd = DL.InputReader(cfg, training, True, device="cpu", seed=9, rank=r, world=2)
            assert rd.local_batch(8) == 4 and rd.local_batch(None) is None
            shards.append(list(rd._records(pattern, rd.local_batch(8))))
        assert not set(shards[0]) & set(shards[1]) and len(shards[0]) == len(shards[1])     # equal shards: no rank runs ahead
        if training:    # complete groups of `world` records: all 14 here (15 would drop one per pass)
            assert sorted(shards[0] + shards[1]) == sorted(recs)
        else:           # evaluation: rank r's quarter-batches of each COMPLETE global batch of 8; the partial one (6) is dropped
            assert shards[0] == one[0:4] and shards[1] == one[4:8]
    # an odd record count in training: the pass's last record is dropped on both ranks (equal shards), every pass
    DL.write_tfrecords(str(tmp_path / "part-9.tfrecord"), [b"\xff" * 40])
    odd = [list(DL.InputReader(cfg, True, True, device="cpu", seed=9, rank=r, world=2)._records(pattern, 4)) for r in range(2)]
    assert len(odd[0]) == len(odd[1]) == 7 and not set(odd[0]) & set(odd[1])
    os.remove(str(tmp_path / "part-9.tfrecord"))
    with pytest.raises(ValueError):
        DL.InputReader(cfg, True, True, device="cpu", rank=0, world=3).local_batch(8)
    with pytest.raises(ValueError):
        DL.InputReader(cfg, True, True, device="cpu", rank=2, world=2)
    # text-file path: the same sharding of the line list
    (tmp_path / "list.txt").write_text("".join(f"v{i}.mp4 {i}\n" for i in range(7)))
    a = list(DL.InputReader(cfg, False, False, device="cpu", rank=0, world=2)._records(str(tmp_path / "list.txt"), None))
    b = list(DL.InputReader(cfg, False, False, device="cpu", rank=1, world=2)._records(str(tmp_path / "list.txt"), None))
    assert a == [f"v{i}.mp4 {i}" for i in (0, 2, 4)] and b == [f"v{i}.mp4 {i}" for i in (1, 3, 5)]     # (v6: incomplete group)
    # training: the WHOLE list is re-permuted every pass by the rank-independent generator before the split, so a rank's
    # shard changes from pass to pass (reference: .shuffle(DATASET_SIZE) of the one dataset, then the replica split)
    ra = DL.InputReader(cfg, True, False, device="cpu", seed=4, rank=0, world=2)
    rb = DL.InputReader(cfg, True, False, device="cpu", seed=4, rank=1, world=2)
    passes = [(list(ra._records(str(tmp_path / "list.txt"), 2)), list(rb._records(str(tmp_path / "list.txt"), 2))) for _ in range(4)]
    for pa, pb in passes:
        assert len(pa) == len(pb) == 3 and not set(pa) & set(pb)
    assert len({tuple(sorted(pa)) for pa, _ in passes}) > 1


def _eval_rank(rank, world, port, pattern, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import x3d_tf_amd as x
    from x3d_tf_amd.evaluate import evaluate_dataset
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = x.get_config("XS")
    res = evaluate_dataset(_FakeModel(), cfg, _fake_batches(DL.InputReader(cfg, False, True, device="cpu"), pattern, 4))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


class _FakeModel:
    """probabilities that depend on the record only (evaluate_dataset's arithmetic and exchange are under test, not X3D)"""
    table = torch.softmax(torch.randn(64, 10, generator=torch.Generator().manual_seed(3)) * 3, 1)

    def __call__(self, clips, training=False):
        return self.table[clips.view(-1).long()]


def _fake_batches(rd, pattern, global_batch):
    """(record ids, labels) batches from the reader's own record stream and batch arithmetic (the GPU view kernels are not
    available on the CPU: the record's first byte stands for the clip)"""
    b = rd.local_batch(global_batch)
    recs = [r[0] for r in rd._records(pattern, b)]
    for k in range(0, len(recs) - b + 1, b):        # drop_remainder
        ids = torch.tensor(recs[k:k + b])
        yield ids.view(-1, 1), ids % 10


def test_two_rank_evaluation_reports_the_single_process_metric(tmp_path):
    """under torchrun every rank evaluated 1/world of the videos and reported a metric over its shard only.
    evaluate_dataset sums the counters over the ranks: both ranks of a gloo world-2 run return what one process computes
    over the whole stream -- the same videos, the same trailing partial batch dropped (reference eval.py:83-89 under
    MirroredStrategy reports ONE global metric)."""
    import torch.multiprocessing as mp
    import x3d_tf_amd as x
    from x3d_tf_amd.evaluate import evaluate_dataset
    from tests.test_dist import _free_port
    recs = [bytes([i]) * (3 + i) for i in range(15)]          # 15 videos, global batch 4: three batches, 3 videos dropped
    for k in range(0, 15, 4):
        DL.write_tfrecords(str(tmp_path / f"part-{k // 4}.tfrecord"), recs[k:k + 4])
    pattern = str(tmp_path / "part-*.tfrecord")
    cfg = x.get_config("XS")
    want = evaluate_dataset(_FakeModel(), cfg, _fake_batches(DL.InputReader(cfg, False, True, device="cpu", rank=0, world=1), pattern, 4))
    assert want["videos"] == 12
    out = str(tmp_path / "r0.pt")
    mp.spawn(_eval_rank, args=(2, _free_port(), pattern, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["videos"] == 12
    for k in ("loss", "acc", "top_5_acc"):
        assert abs(got[k] - want[k]) < 1e-12, (k, got[k], want[k])


# ---- the pipeline end to end on the GPU ------------------------------------------------------------
def _write_dataset(tmp_path, vids, per_file=3):
    paths = []
    for k in range(0, len(vids), per_file):
        p = str(tmp_path / f"kinetics-val-{k // per_file}-of-x.tfrecord")
        DL.write_tfrecords(p, [DL.make_sequence_example(v, lab) for v, lab in vids[k:k + per_file]])
        paths.append(p)
    return str(tmp_path / "kinetics-val-*.tfrecord"), paths


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_pipeline_matches_oracle(gpu, tmp_path, dtype):
    """TFRecords -> parse -> JPEG decode -> x3d_eval_views -> batches: bit-identical to the oracle's view construction
    on the same decoded frames, deterministic order, trailing partial batch dropped (dataloader.py:186)."""
    import x3d_tf_amd as x
    from oracle import views_oracle as V
    cfg = x.get_config("XS", ["DATA.TEMP_DURATION", 4, "DATA.TEST_CROP_SIZE", 16, "TEST.NUM_TEMPORAL_VIEWS", 2,
                              "TEST.NUM_SPATIAL_CROPS", 3])
    vids = _videos(5, seed=1)
    pattern, paths = _write_dataset(tmp_path, vids, per_file=3)
    # interleave order: file 0 record 0, file 1 record 0, file 0 record 1, file 1 record 1, file 0 record 2
    order = [0, 3, 1, 4, 2]
    decoded = {i: DL.InputReader.parse_and_decode(None, DL.make_sequence_example(*vids[i])) for i in range(5)}
    rd = DL.InputReader(cfg, False, True, mixed_precision=dtype != torch.float32, device=gpu, dtype=dtype)
    batches = list(rd(pattern, 2))
    assert len(batches) == 2                                     # 5 videos, batch 2, drop_remainder
    for b, (clips, labels) in enumerate(batches):
        assert clips.is_cuda and clips.dtype == dtype and tuple(clips.shape) == (2 * 6, 4, 16, 16, 3)
        for j in range(2):
            i = order[2 * b + j]
            ref = V.eval_views(decoded[i][0], 4, 2, 3, 16, cfg.DATA.MEAN, cfg.DATA.STD)
            want = torch.from_numpy(np.ascontiguousarray(ref)).to(dtype)
            assert torch.equal(clips[6 * j:6 * j + 6].cpu(), want), (b, j)
            assert int(labels[j]) == vids[i][1]


@pytest.mark.gpu
def test_evaluate_dataset_equals_evaluate_on_decoded_videos(gpu, tmp_path):
    """eval.py:75-89: the metrics over InputReader batches equal the metrics of `evaluate` on the same decoded videos."""
    import x3d_tf_amd as x
    from x3d_tf_amd.evaluate import evaluate, evaluate_dataset
    cfg = x.get_config("XS", ["DATA.TEMP_DURATION", 4, "DATA.TEST_CROP_SIZE", 32, "TEST.NUM_TEMPORAL_VIEWS", 2,
                              "TEST.NUM_SPATIAL_CROPS", 3, "TEST.BATCH_SIZE", 2, "NETWORK.NUM_CLASSES", 10])
    vids = [(v, i % 10) for i, (v, _) in enumerate(_videos(4, seed=6, h=40, w=52))]
    pattern, _ = _write_dataset(tmp_path, vids, per_file=2)
    model = x.X3D(cfg, dtype=torch.float32, device=gpu, seed=3)
    got = evaluate_dataset(model, cfg, DL.InputReader(cfg, False, True, device=gpu)(pattern, cfg.TEST.BATCH_SIZE))
    decoded = [(torch.from_numpy(DL.InputReader.parse_and_decode(None, DL.make_sequence_example(v, lab))[0]).to(gpu), lab)
               for v, lab in vids]
    want = evaluate(model, cfg, decoded)
    assert got["videos"] == want["videos"] == 4
    for k in ("loss", "acc", "top_5_acc"):
        assert abs(got[k] - want[k]) < 1e-6, k


@pytest.mark.gpu
def test_train_pipeline_matches_oracle(gpu, tmp_path):
    """Training mode: shuffled, repeated, one augmented clip per video; given the draws the reader made, every clip of a
    batch is bit-identical to the oracle's train_clip; one epoch sees every video exactly once."""
    import x3d_tf_amd as x
    from oracle import views_oracle as V
    cfg = x.get_config("XS", ["DATA.TEMP_DURATION", 4, "DATA.TRAIN_CROP_SIZE", 16, "DATA.TRAIN_JITTER_SCALES", [18, 22],
                              "DATA.FRAME_RATE", 2])
    vids = _videos(6, seed=2)
    by_label = {}
    for v, lab in vids:
        by_label[lab] = DL.InputReader.parse_and_decode(None, DL.make_sequence_example(v, lab))[0]
    assert len(by_label) == 6
    pattern, _ = _write_dataset(tmp_path, vids, per_file=2)
    rd = DL.InputReader(cfg, True, True, device=gpu, seed=11, num_workers=2)
    it = rd(pattern, 3)
    seen = []
    for step in range(4):                                        # two epochs of 6 videos in batches of 3
        clips, labels = next(it)
        assert tuple(clips.shape) == (3, 4, 16, 16, 3) and clips.dtype == torch.float32
        params = list(rd.last_params)
        assert len(params) == 3 and all(p["flip"] for p in params)   # every training clip is mirrored (transforms.py:205-206)
        for j in range(3):
            lab = int(labels[j])
            p = params[j]
            ref = V.train_clip(by_label[lab], 4, 2, p["start"], p["jitter"], 16, p["y0"], p["x0"], True, cfg.DATA.MEAN, cfg.DATA.STD)
            assert torch.equal(clips[j].cpu(), torch.from_numpy(np.ascontiguousarray(ref))), (step, j)
            seen.append(lab)
    it.close()
    assert sorted(seen[:6]) == sorted(by_label) and sorted(seen[6:]) == sorted(by_label)   # repeat(): epoch after epoch
    assert seen[:6] != [lab for _, lab in vids]                                            # shuffled


@pytest.mark.gpu
def test_fit_from_tfrecords(gpu, tmp_path):
    """reference train.py:128-152 end to end on synthetic records: InputReader -> Trainer.fit (per-epoch lr_schedule,
    ckpt-<epoch> after every epoch) -> resume finds the last checkpoint."""
    import x3d_tf_amd as x
    from x3d_tf_amd.model import X3D
    from x3d_tf_amd.train import Trainer
    cfg = x.get_config("XS", ["DATA.TEMP_DURATION", 4, "DATA.TRAIN_CROP_SIZE", 32, "DATA.TRAIN_JITTER_SCALES", [34, 40],
                              "DATA.FRAME_RATE", 1, "NETWORK.NUM_CLASSES", 5, "TRAIN.BATCH_SIZE", 2, "TRAIN.DATASET_SIZE", 4,
                              "TRAIN.EPOCHS", 2])
    vids = [(v, i % 5) for i, (v, _) in enumerate(_videos(4, seed=4, h=40, w=48))]
    pattern, _ = _write_dataset(tmp_path, vids, per_file=2)
    m = X3D(cfg, dtype=torch.float32, device=gpu, seed=1)
    tr = Trainer(m, cfg)
    w0 = m.flat_params.clone()
    ds = DL.InputReader(cfg, True, True, device=gpu, seed=3)(pattern, cfg.TRAIN.BATCH_SIZE)
    hist = tr.fit(ds, model_dir=str(tmp_path / "run"))
    ds.close()
    assert len(hist) == 2 and all(np.isfinite(h) for h in hist) and tr.epoch == 2 and tr.opt_step == 4
    assert not torch.equal(w0, m.flat_params)
    assert sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "run" / "ckpt-*.index"))) == ["ckpt-1.index", "ckpt-2.index"]
    tr2 = Trainer(X3D(cfg, dtype=torch.float32, device=gpu, seed=9), cfg)
    assert tr2.resume(str(tmp_path / "run")) == 2
