"""TF tensor-bundle reader/writer against the manifest of the reference's real checkpoint index
(tests/golden/model_index_manifest.json, extracted from models/X3D-M/model.index by make_golden.py)."""
import json
import os

import pytest
import torch

import x3d_tf_amd as x
from x3d_tf_amd import arch as A
from x3d_tf_amd import checkpoint as ck
from x3d_tf_amd.params import init_params, randomize_bn_

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(GOLDEN, "model_index_manifest.json")))


def _specs(variant="M"):
    arch = x.build_arch(x.get_config(variant))
    return arch, {s.name: s for s in A.param_specs(arch)}


def test_variable_names_and_shapes_match_the_released_checkpoint():
    arch, specs = _specs("M")
    ent = {e["key"]: e for e in MAN["entries"]}
    assert MAN["identical_layout_XS_S_M"] is True and len(ent) == 789
    model_keys = {n + ck.SUFFIX for n in specs}
    assert len(model_keys) == 476 and model_keys <= set(ent)
    total = 0
    for n, s in specs.items():
        e = ent[n + ck.SUFFIX]
        assert e["dtype"] == ck.DT_FLOAT and tuple(e["shape"]) == ck.tf_shape(s), n
        numel = 1
        for d in s.shape:
            numel *= d
        assert e["size"] == 4 * numel
        total += numel
    assert total == 3795830
    # everything else in the bundle is optimizer state or the object graph
    rest = [k for k in ent if k not in model_keys]
    slots = [k for k in rest if "/.OPTIMIZER_SLOT/optimizer/momentum/" in k]
    assert len(slots) == 308 and len(rest) == 308 + 5


def test_crc32c_known_answers():
    assert ck.crc32c(b"123456789") == 0xE3069283
    assert ck.crc32c(b"") == 0
    assert ck.crc32c(b"a" * 1000, 0) == ck.crc32c(b"a" * 400, ck.crc32c(b"a" * 600))   # the crc argument chains
    assert ck.mask_crc(0xE3069283) == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_bundle_roundtrip_with_reference_layout(tmp_path):
    """write a synthetic bundle laid out with the reference's keys, read it back through the index parser"""
    arch, specs = _specs("M")
    state = randomize_bn_(init_params(arch, seed=5), seed=6)
    mom = {k: torch.randn_like(v) for k, v in state.items() if specs[k].trainable}
    prefix = str(tmp_path / "model")
    ck.write_checkpoint(prefix, state, specs, momentum=mom)
    hdr, ent = ck.read_index(prefix + ".index")
    assert hdr["num_shards"] == 1
    gold = {e["key"]: e for e in MAN["entries"]}
    assert set(ent) == set(gold)                  # all 789 keys of the released file: variables, momentum slots, optimizer
    for k, e in ent.items():                      # hyper-parameters, object graph -- with the same shapes / sizes
        assert list(e.shape) == gold[k]["shape"], k
        if k != ck.OBJECT_GRAPH_KEY:              # (Keras' own graph carries bookkeeping edges this writer omits)
            assert e.size == gold[k]["size"], k
    assert ent[ck.OBJECT_GRAPH_KEY].dtype == ck.DT_STRING and ent["optimizer/iter" + ck.SUFFIX].dtype == ck.DT_INT64
    # the object graph: following the `children` edges named by a key's segments from the root reaches a variable node
    # whose VARIABLE_VALUE attribute carries exactly that checkpoint key (how Keras' object-based load_weights finds it);
    # every momentum slot hangs off the optimizer node and points at its variable
    nodes = ck.read_object_graph(prefix)

    def walk(path):
        cur = 0
        for seg in path.split("/"):
            cur = nodes[cur]["children"][seg]
        return cur
    for name in specs:
        assert nodes[walk(name)]["attributes"] == {"VARIABLE_VALUE": name + ck.SUFFIX}, name
    for h in ck.OPTIMIZER_HYPER:
        assert nodes[walk("optimizer/" + h)]["attributes"]["VARIABLE_VALUE"] == f"optimizer/{h}{ck.SUFFIX}"
    slots = nodes[walk("optimizer")]["slots"]
    assert len(slots) == len(mom) == 308
    for var_node, slot_name, slot_node in slots:
        vkey = nodes[var_node]["attributes"]["VARIABLE_VALUE"]
        assert slot_name == "momentum"
        assert nodes[slot_node]["attributes"]["VARIABLE_VALUE"] == vkey[:-len(ck.SUFFIX)] + "/.OPTIMIZER_SLOT/optimizer/momentum" + ck.SUFFIX
    assert len(nodes) == len({id(n) for n in nodes}) and all(0 <= c < len(nodes) for n in nodes for c in n["children"].values())
    back, mom_back = ck.read_checkpoint(str(tmp_path), specs, with_momentum=True)   # directory -> `checkpoint` file
    for k in state:
        assert torch.equal(back[k], state[k]), k
    for k in mom:
        assert torch.equal(mom_back[k], mom[k]), k
    # layouts: TF kernel [kt,kh,kw,Cin/g,Cout]
    s = specs["conv1/conv_s/kernel"]
    tfk = ck.to_tf(s, state["conv1/conv_s/kernel"])
    assert tuple(tfk.shape) == (1, 3, 3, 3, 24) and tfk[0, 1, 2, 0, 5] == state["conv1/conv_s/kernel"][5, 0, 1, 2]
    s = specs["stages/0/stage/layer_with_weights-0/bottleneck/b/kernel"]
    tfk = ck.to_tf(s, state[s.name])
    assert tuple(tfk.shape) == (3, 3, 3, 1, 54) and tfk[2, 0, 1, 0, 7] == state[s.name][7, 2, 0, 1]
    tfk = ck.to_tf(specs["fc2/kernel"], state["fc2/kernel"])
    assert tuple(tfk.shape) == (2048, 400)


def test_adam_bundle_roundtrip(tmp_path):
    """the optimizer branch train.py:93-95 saved by ModelCheckpoint (utils.py:128-132): Keras Adam's slot variables `m` /
    `v` per trainable variable and its hyper variables (iter, learning_rate, decay, beta_1, beta_2) round trip; no SGD
    `momentum` key is written, and the object graph names both slot kinds."""
    arch, specs = _specs("XS")
    state = randomize_bn_(init_params(arch, seed=5), seed=6)
    m1 = {k: torch.randn_like(v) for k, v in state.items() if specs[k].trainable}
    m2 = {k: torch.rand_like(v) for k, v in state.items() if specs[k].trainable}
    prefix = str(tmp_path / "ckpt-2")
    ck.write_checkpoint(prefix, state, specs, slots={"m": m1, "v": m2},
                        optimizer_hyper=dict(iter=17, learning_rate=0.25, beta_1=0.9, beta_2=0.999, decay=0.0))
    _, ent = ck.read_index(prefix + ".index")
    assert not [k for k in ent if "/optimizer/momentum/" in k or k.startswith("optimizer/momentum")]
    assert {k[len("optimizer/"):-len(ck.SUFFIX)] for k in ent if k.startswith("optimizer/")} == set(ck.ADAM_HYPER)
    assert sum("/.OPTIMIZER_SLOT/optimizer/m/" in k for k in ent) == len(m1)
    assert sum("/.OPTIMIZER_SLOT/optimizer/v/" in k for k in ent) == len(m2)
    back, slots, hyper = ck.read_checkpoint(prefix, specs, with_slots=("momentum", "m", "v"))
    assert hyper["iter"] == 17 and abs(hyper["learning_rate"] - 0.25) < 1e-7 and abs(hyper["beta_2"] - 0.999) < 1e-7
    assert not slots["momentum"]
    for k in m1:
        assert torch.equal(slots["m"][k], m1[k]) and torch.equal(slots["v"][k], m2[k]), k
    for k in state:
        assert torch.equal(back[k], state[k]), k
    nodes = ck.read_object_graph(prefix)
    opt = nodes[nodes[0]["children"]["optimizer"]]
    kinds = {}
    for var_node, slot_name, slot_node in opt["slots"]:
        vkey = nodes[var_node]["attributes"]["VARIABLE_VALUE"]
        assert nodes[slot_node]["attributes"]["VARIABLE_VALUE"] == \
            vkey[:-len(ck.SUFFIX)] + f"/.OPTIMIZER_SLOT/optimizer/{slot_name}" + ck.SUFFIX
        kinds[slot_name] = kinds.get(slot_name, 0) + 1
    assert kinds == {"m": len(m1), "v": len(m2)}
    assert set(opt["children"]) == set(ck.ADAM_HYPER)


def test_corruption_and_partial_are_detected(tmp_path):
    arch, specs = _specs("XS")
    state = init_params(arch, seed=1)
    prefix = str(tmp_path / "model")
    ck.write_checkpoint(prefix, state, specs)
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    _, ent = ck.read_index(prefix + ".index")
    raw[ent["fc2/kernel" + ck.SUFFIX].offset + 100] ^= 0xFF          # inside a model variable
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="CRC32C"):
        ck.read_checkpoint(prefix, specs)
    raw[ent["fc2/kernel" + ck.SUFFIX].offset + 100] ^= 0xFF
    raw[ent[ck.OBJECT_GRAPH_KEY].offset + 1000] ^= 0xFF              # inside the object graph
    open(data, "wb").write(bytes(raw))
    ck.read_checkpoint(prefix, specs)                                # the variables are intact ...
    with pytest.raises(ValueError, match="checksum"):
        ck.read_object_graph(prefix)                                 # ... the graph's checksum is not
    os.remove(data)
    with pytest.raises(FileNotFoundError, match="data shard"):
        ck.read_checkpoint(prefix, specs)
    # a model variable absent from the bundle always raises
    ck.write_checkpoint(prefix, state, {k: v for k, v in specs.items() if k != "fc2/bias"})
    with pytest.raises(KeyError):
        ck.read_checkpoint(prefix, specs)
    # an extra key raises only without expect_partial
    ck.write_checkpoint(prefix, state, specs)
    sub = {k: v for k, v in specs.items() if k != "fc2/bias"}
    ck.read_checkpoint(prefix, sub, expect_partial=True)
    with pytest.raises(KeyError):
        ck.read_checkpoint(prefix, sub, expect_partial=False)
    assert ck.latest_checkpoint(str(tmp_path)) == prefix and ck.latest_checkpoint(str(tmp_path / "nope")) is None
