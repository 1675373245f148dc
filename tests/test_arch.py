"""Structural pins: the only oracles the reference ships are its Keras summaries (models/X3D-*/X3D_*.txt,
committed as tests/golden/summaries.json) and its checkpoint indexes.  Every output shape and parameter
count of all five variants must be reproduced exactly."""
import json
import os

import pytest

import x3d_tf_amd as x
from x3d_tf_amd import arch as A

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUM = json.load(open(os.path.join(GOLDEN, "summaries.json")))


@pytest.mark.parametrize("variant", ["XS", "S", "M", "L", "XL"])
def test_summary_matches_reference(variant):
    gold = SUM[variant]
    cfg = x.get_config(variant)
    arch = x.build_arch(cfg)
    t, h, w, c = gold["rows"][0]["shape"]
    assert (t, h) == (cfg.DATA.TEMP_DURATION, cfg.DATA.TEST_CROP_SIZE) and c == 3
    rows = A.summary_rows(arch, t, h, w, c)
    assert [r[0] for r in rows] == [g["name"] for g in gold["rows"][1:]]
    for (name, shape, params), g in zip(rows, gold["rows"][1:]):
        assert list(shape) == g["shape"], name
        assert params == g["params"], name
    tot, tr, ntr = A.count_params(arch)
    assert (tot, tr, ntr) == (gold["totals"]["total"], gold["totals"]["trainable"], gold["totals"]["non-trainable"])


def test_round_width_and_repeats():
    # reference utils.py:7-40
    assert A.round_width(12, 2) == 24 and A.round_width(24, 2) == 48 and A.round_width(24, 8) == 192
    assert A.round_width(12, 2.9) == 32 and A.round_width(54, 0.0625) == 8 and A.round_width(432, 0.0625) == 32
    assert A.round_width(630, 0.0625) == 40 and A.round_width(17, 0) == 17 and A.round_width(8, 0.1) == 8
    assert A.round_width(10, 1.25) == 16      # 12.5 -> 8 is < 0.9*12.5 -> +8
    assert A.round_repeats(5, 2.2) == 11 and A.round_repeats(3, 5.0) == 15 and A.round_repeats(3, 0) == 3


def test_se_placement_is_global_odd_blocks():
    # the released checkpoints hold SE weights for stage0 {0,2}, stage1 {1,3},
    # stage2 {0,2,4,6,8,10}, stage3 {1,3,5}
    arch = x.build_arch(x.get_config("M"))
    se = {}
    for b in arch.blocks:
        if b.has_se:
            se.setdefault(b.stage, []).append(b.index)
    assert se == {0: [0, 2], 1: [1, 3], 2: [0, 2, 4, 6, 8, 10], 3: [1, 3, 5]}
    assert sum(b.has_se for b in x.build_arch(x.get_config("L")).blocks) == 28
    assert [b.se_width for b in arch.blocks if b.has_se][:3] == [8, 8, 8]
    # a second model in the same process starts counting from 1 again (explicit per-model counter)
    arch2 = x.build_arch(x.get_config("L"))
    assert arch2.blocks[0].has_se and arch2.blocks[0].global_index == 1


def test_first_block_of_every_stage_has_conv_shortcut():
    arch = x.build_arch(x.get_config("M"))
    for st in arch.stages:
        assert st.blocks[0].has_shortcut_conv and st.blocks[0].stride == 2
        assert all(not b.has_shortcut_conv and b.stride == 1 for b in st.blocks[1:])
    assert arch.stages[0].blocks[0].cin == arch.stages[0].blocks[0].cout == 24   # 24 -> 24 still gets one


def test_same_padding_rule():
    assert A.same_pad(112, 3, 2) == (56, 0, 1) and A.same_pad(39, 3, 2) == (20, 1, 1)
    assert A.same_pad(7, 3, 1) == (7, 1, 1) and A.same_pad(16, 3, 1) == (16, 1, 1)


def test_workload_accounting_matches_baseline():
    # X3D-M 210.3 M elements and 9.465 GFLOP per clip forward
    arch = x.build_arch(x.get_config("M"))
    w = A.workload(arch, 16, 224, 224)
    assert abs(w["total_elements"] / 1e6 - 210.3) < 0.1
    assert abs(w["total_flops"] / 1e9 - 9.465) < 0.01
    assert abs(w["elements"]["depthwise"] / 1e6 - 74.3) < 0.1 and abs(w["elements"]["tail"] / 1e6 - 33.0) < 0.1


def test_config_schema():
    cfg = x.get_config("M")
    assert cfg.NETWORK.BN.EPS == 1e-5 and cfg.NETWORK.WEIGHT_DECAY == 5e-5 and cfg.NETWORK.DROPOUT_RATE == 0.5
    assert cfg.TEST.NUM_TEMPORAL_VIEWS == 10 and cfg.TEST.NUM_SPATIAL_CROPS == 1 and cfg.TRAIN.OPTIMIZER == "sgd"
    with pytest.raises(AttributeError):
        cfg.NETWORK.WIDTH_FACTOR = 2.0       # frozen
    d = x.get_default_config()
    assert d.NETWORK.C1_CHANNELS == 12 and d.TEST.NUM_SPATIAL_CROPS == 3 and d.NETWORK.SCALE_RES2 is False
    with pytest.raises(KeyError):
        d.merge_from_list(["NETWORK.NOT_A_KEY", 1])
    xl = x.get_config("XL")
    assert xl.NETWORK.SCALE_RES2 is True and xl.NETWORK.WIDTH_FACTOR == 2.9
    assert dict(xl)["TRAIN"]["BATCH_SIZE"] == 16
