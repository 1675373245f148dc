"""Reader / writer for the reference's checkpoints: TensorFlow "tensor bundle" files
(``<prefix>.index`` = LevelDB-style SSTable of BundleEntryProto, ``<prefix>.data-00000-of-00001`` = raw
little-endian tensors), as released under reference models/X3D-{XS,S,M}/ and as written/read by
``model.save_weights`` / ``model.load_weights`` in reference train.py:131-143 and eval.py:78-81.

TensorFlow is not needed (and not available): the container formats are restated here.
  - SSTable: 48-byte footer (metaindex handle, index handle, padding, magic 0xdb4775248b80fb57), blocks of
    prefix-compressed (shared, non_shared, value_len, key_delta, value) records followed by a restart
    array, each block trailed by a 1-byte compression type and a masked CRC32C.
  - BundleEntryProto: dtype(1) shape(2) shard_id(3) offset(4) size(5) crc32c(6, fixed32, masked).
Variable keys are the Keras object-graph paths of reference model.py's attributes with the suffix
``/.ATTRIBUTES/VARIABLE_VALUE``; those paths ARE this build's parameter names.
"""
import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_STRING, DT_INT64 = 1, 7, 9
_NP = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 14: None}  # 14 = bfloat16 (unused here)


# ------------------------------------------------------------------------------------------------
# CRC32C (Castagnoli), masked as in LevelDB / TF:  ((crc >> 15) | (crc << 17)) + 0xa282ead8
# ------------------------------------------------------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        tbl = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            tbl[i] = c
        _CRC_TABLE = tbl
    return _CRC_TABLE


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC32C of `data`.  Uses the native helper in libx3d_hip.so when it is loadable (15 MB checkpoints),
    a table-driven Python loop otherwise."""
    try:
        from . import hip
        return int(hip.load().x3d_crc32c(bytes(data), len(data), crc))
    except Exception:
        tbl = _crc_table()
        c = crc ^ 0xFFFFFFFF
        for b in data:
            c = int(tbl[(c ^ b) & 0xFF]) ^ (c >> 8)
        return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# varints / minimal protobuf
# ------------------------------------------------------------------------------------------------
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """field number -> list of raw values (int for varint/fixed, bytes for length-delimited)."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.setdefault(field, []).append(v)
    return out


@dataclass
class BundleEntry:
    dtype: int
    shape: Tuple[int, ...]
    shard_id: int
    offset: int
    size: int
    crc32c: int


def _parse_entry(value: bytes) -> BundleEntry:
    f = _parse_proto(value)
    shape = []
    if 2 in f:
        for dim in _parse_proto(f[2][0]).get(2, []):
            d = _parse_proto(dim)
            shape.append(d.get(1, [0])[0])
    return BundleEntry(dtype=f.get(1, [0])[0], shape=tuple(shape), shard_id=f.get(3, [0])[0],
                       offset=f.get(4, [0])[0], size=f.get(5, [0])[0], crc32c=f.get(6, [0])[0])


def _encode_entry(e: BundleEntry) -> bytes:
    out = bytearray()
    out += _put_varint((1 << 3) | 0) + _put_varint(e.dtype)
    shp = bytearray()
    for d in e.shape:
        dim = _put_varint((1 << 3) | 0) + _put_varint(d)
        shp += _put_varint((2 << 3) | 2) + _put_varint(len(dim)) + dim
    out += _put_varint((2 << 3) | 2) + _put_varint(len(shp)) + bytes(shp)
    if e.shard_id:
        out += _put_varint((3 << 3) | 0) + _put_varint(e.shard_id)
    if e.offset:
        out += _put_varint((4 << 3) | 0) + _put_varint(e.offset)
    out += _put_varint((5 << 3) | 0) + _put_varint(e.size)
    out += _put_varint((6 << 3) | 5) + struct.pack("<I", e.crc32c)
    return bytes(out)


# ------------------------------------------------------------------------------------------------
# SSTable
# ------------------------------------------------------------------------------------------------
def _read_block(data: bytes, off: int, size: int, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    blk = data[off:off + size]
    ctype = data[off + size]
    if ctype != 0:
        raise ValueError("compressed SSTable blocks are not supported (TF writes bundle indexes uncompressed)")
    if verify:
        stored = struct.unpack_from("<I", data, off + size + 1)[0]
        if mask_crc(crc32c(data[off:off + size + 1])) != stored:
            raise ValueError("SSTable block checksum mismatch")
    n_restarts = struct.unpack_from("<I", blk, len(blk) - 4)[0]
    end = len(blk) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(blk, pos)
        non_shared, pos = _get_varint(blk, pos)
        vlen, pos = _get_varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(index_path: str, verify: bool = True) -> Tuple[dict, Dict[str, BundleEntry]]:
    """Parse ``<prefix>.index`` -> (header fields, {key: BundleEntry}) in key order."""
    data = open(index_path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
        raise ValueError(f"{index_path}: not a TF tensor-bundle index (bad magic)")
    footer = data[-48:]
    pos = 0
    _, pos = _get_varint(footer, pos)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries: Dict[str, BundleEntry] = {}
    header = {}
    for _, handle in _read_block(data, ioff, isize, verify):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, value in _read_block(data, boff, bsize, verify):
            if key == b"":
                h = _parse_proto(value)
                header = dict(num_shards=h.get(1, [1])[0], endianness=h.get(2, [0])[0])
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    return header, entries


def _build_block(records: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts = []
    prev = b""
    for i, (k, v) in enumerate(records):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_index(index_path: str, entries: Dict[str, BundleEntry], block_size: int = 4096):
    """Write a bundle index readable by read_index (and by TensorFlow's BundleReader)."""
    header = (_put_varint((1 << 3) | 0) + _put_varint(1) +            # num_shards = 1
              _put_varint((3 << 3) | 2) + _put_varint(2) + _put_varint((1 << 3) | 0) + _put_varint(1))  # version.producer = 1
    records = [(b"", header)] + [(k.encode("utf-8"), _encode_entry(e)) for k, e in sorted(entries.items())]
    out = bytearray()
    index_records = []

    def emit(block: bytes) -> Tuple[int, int]:
        off = len(out)
        out.extend(block)
        out.append(0)
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)

    cur: List[Tuple[bytes, bytes]] = []
    cur_size = 0
    for k, v in records:
        cur.append((k, v))
        cur_size += len(k) + len(v) + 3
        if cur_size >= block_size:
            off, size = emit(_build_block(cur))
            index_records.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
            cur, cur_size = [], 0
    if cur:
        off, size = emit(_build_block(cur))
        index_records.append((cur[-1][0], _put_varint(off) + _put_varint(size)))
    moff, msize = emit(_build_block([]))
    ioff, isize = emit(_build_block(index_records, restart_interval=1))
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    footer = footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    out.extend(footer)
    with open(index_path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------------------------
# layouts: TF kernel [k..., Cin/groups, Cout] <-> native (x3d-tf_amd/arch.py)
# ------------------------------------------------------------------------------------------------
def tf_shape(spec) -> Tuple[int, ...]:
    s = spec.shape
    if spec.kind == "pw":
        return (1, 1, 1, s[1], s[0])
    if spec.kind == "dw":
        return (3, 3, 3, 1, s[0])
    if spec.kind == "stem_s":
        return (1, 3, 3, s[1], s[0])
    if spec.kind == "stem_t":
        return (s[1], 1, 1, 1, s[0])
    if spec.kind == "dense":
        return (s[1], s[0])
    return tuple(s)


def to_native(spec, t: torch.Tensor) -> torch.Tensor:
    if spec.kind in ("pw", "dw", "stem_s", "stem_t"):
        return t.permute(4, 3, 0, 1, 2).reshape(spec.shape).contiguous()
    if spec.kind == "dense":
        return t.t().contiguous()
    return t.reshape(spec.shape).contiguous()


def to_tf(spec, t: torch.Tensor) -> torch.Tensor:
    s = spec.shape
    if spec.kind == "pw":
        return t.reshape(s[0], s[1], 1, 1, 1).permute(2, 3, 4, 1, 0).contiguous()
    if spec.kind == "dw":
        return t.reshape(s[0], 1, 3, 3, 3).permute(2, 3, 4, 1, 0).contiguous()
    if spec.kind == "stem_s":
        return t.reshape(s[0], s[1], 1, 3, 3).permute(2, 3, 4, 1, 0).contiguous()
    if spec.kind == "stem_t":
        return t.reshape(s[0], 1, s[1], 1, 1).permute(2, 3, 4, 1, 0).contiguous()
    if spec.kind == "dense":
        return t.t().contiguous()
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# checkpoint level
# ------------------------------------------------------------------------------------------------
def latest_checkpoint(directory: str) -> Optional[str]:
    """tf.train.latest_checkpoint: read the ``checkpoint`` state file (reference models/X3D-M/checkpoint:1-2)."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    for line in open(state):
        if line.startswith("model_checkpoint_path:"):
            name = line.split(":", 1)[1].strip().strip('"')
            prefix = name if os.path.isabs(name) else os.path.join(directory, name)
            return prefix if os.path.exists(prefix + ".index") else None
    return None


def resolve_prefix(path: str) -> str:
    if os.path.isdir(path):
        p = latest_checkpoint(path)
        if p is None:
            raise FileNotFoundError(f"no checkpoint found in {path}")
        return p
    if path.endswith(".index"):
        path = path[:-6]
    if not os.path.exists(path + ".index"):
        raise FileNotFoundError(path + ".index")
    return path


def read_checkpoint(path: str, specs: Dict[str, object], expect_partial: bool = True, verify_crc: bool = True,
                    with_momentum: bool = False, with_slots: Tuple[str, ...] = ()):
    """Read every model variable named in `specs` from a bundle -> {native name: fp32 CPU tensor}.

    Keys of the bundle that the model does not own (optimizer iter/lr/momentum slots, the object graph) are
    ignored when expect_partial (eval.py:81 ``.expect_partial()``); model variables missing from the bundle
    always raise.  with_momentum: also return the SGD momentum slots ({name: tensor}).  with_slots=("momentum", "m",
    "v"): return (variables, {slot kind: {name: tensor}}, {optimizer hyper-parameter: value}) -- everything the
    reference's `model.load_weights(latest)` (train.py:131-136) restores of the optimizer."""
    prefix = resolve_prefix(path)
    header, entries = read_index(prefix + ".index")
    n_shards = header.get("num_shards", 1)
    shards = {}

    def shard(i):
        if i not in shards:
            fn = f"{prefix}.data-{i:05d}-of-{n_shards:05d}"
            if not os.path.exists(fn):
                raise FileNotFoundError(
                    f"{fn}: the tensor data shard is missing (the reference repository ships only the .index "
                    "files; obtain the shard from the reference's release)")
            shards[i] = np.memmap(fn, dtype=np.uint8, mode="r")
        return shards[i]

    def fetch(key, spec):
        e = entries[key]
        if e.dtype != DT_FLOAT:
            raise ValueError(f"{key}: dtype {e.dtype} is not float32")
        if tuple(e.shape) != tf_shape(spec):
            raise ValueError(f"{key}: bundle shape {tuple(e.shape)} vs model {tf_shape(spec)}")
        raw = bytes(shard(e.shard_id)[e.offset:e.offset + e.size])
        if verify_crc and mask_crc(crc32c(raw)) != e.crc32c:
            raise ValueError(f"{key}: CRC32C mismatch")
        t = torch.from_numpy(np.frombuffer(raw, dtype="<f4").copy()).reshape(e.shape if e.shape else ())
        return to_native(spec, t)

    out, missing = {}, []
    for name, spec in specs.items():
        key = name + SUFFIX
        if key not in entries:
            missing.append(name)
            continue
        out[name] = fetch(key, spec)
    if missing:
        raise KeyError(f"{len(missing)} model variables missing from {prefix}: {missing[:4]}...")
    owned = {n + SUFFIX for n in specs}
    slot_kinds = tuple(with_slots) + (("momentum",) if with_momentum and "momentum" not in with_slots else ())
    for sk in slot_kinds:   # train.py:135 loads the optimizer slots too
        owned |= {f"{n}/.OPTIMIZER_SLOT/optimizer/{sk}{SUFFIX}" for n in specs}
    extra = [k for k in entries if k not in owned]
    if extra and not expect_partial:
        raise KeyError(f"{len(extra)} checkpoint keys unused by the model: {extra[:4]}...")
    slots = {sk: {} for sk in slot_kinds}
    for sk in slot_kinds:
        for name, spec in specs.items():
            k = f"{name}/.OPTIMIZER_SLOT/optimizer/{sk}{SUFFIX}"
            if k in entries:
                slots[sk][name] = fetch(k, spec)
    if with_slots:
        hyper = {}
        for k, e in entries.items():      # optimizer/<hyper>/.ATTRIBUTES/VARIABLE_VALUE scalars (iter int64, the rest float32)
            if k.startswith("optimizer/") and k.endswith(SUFFIX) and not e.shape:
                raw = bytes(shard(e.shard_id)[e.offset:e.offset + e.size])
                if verify_crc and mask_crc(crc32c(raw)) != e.crc32c:
                    raise ValueError(f"{k}: CRC32C mismatch")
                h = k[len("optimizer/"):-len(SUFFIX)]
                if e.dtype == DT_INT64:
                    hyper[h] = struct.unpack("<q", raw)[0]
                elif e.dtype == DT_FLOAT:
                    hyper[h] = struct.unpack("<f", raw)[0]
        return out, slots, hyper
    if with_momentum:
        return out, slots["momentum"]
    return out


# ------------------------------------------------------------------------------------------------
# the object graph Keras restores by (tensorflow/core/protobuf/trackable_object_graph.proto) [TF-3p]
#   TrackableObjectGraph { repeated TrackableObject nodes = 1; }
#   TrackableObject { repeated ObjectReference children = 1; repeated SerializedTensor attributes = 2;
#                     repeated SlotVariableReference slot_variables = 3; }
#   ObjectReference { int32 node_id = 1; string local_name = 2; }
#   SerializedTensor { string name = 1; string full_name = 2; string checkpoint_key = 3; }
#   SlotVariableReference { int32 original_variable_node_id = 1; string slot_name = 2; int32 slot_variable_node_id = 3; }
# `model.load_weights(<TF-format checkpoint>)` (reference train.py:135, eval.py:81) walks the Python object tree from the
# model and follows the proto's `children` edges by local name; a variable is restored from the `checkpoint_key` of its
# VARIABLE_VALUE attribute.  The edges a restore walks are exactly the segments of the variable keys
# ("stages/0/stage/layer_with_weights-0/bottleneck/a/kernel"), so the graph is rebuilt from the keys: one node per key
# prefix.  Keras' own file holds more edges (layer-N aliases, keras_api bookkeeping): they are not needed to restore and
# are not emitted.  The reference's file could not be decoded for comparison (its data shard is not in the checkout).
# ------------------------------------------------------------------------------------------------
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
OPTIMIZER_HYPER = ("decay", "iter", "learning_rate", "momentum")   # optimizer/<name>/.ATTRIBUTES/VARIABLE_VALUE in the released bundles
# tf.keras.optimizers.Adam (OptimizerV2, TF 2.4): hyper variables set with _set_hyper -- learning_rate, decay, beta_1, beta_2
# (epsilon and amsgrad are Python attributes, not variables) -- plus `iter`; slot variables "m" and "v" per trainable
# variable.  [TF-3p]: the reference ships no Adam checkpoint to compare with.
ADAM_HYPER = ("beta_1", "beta_2", "decay", "iter", "learning_rate")


def _pb_len(field: int, payload: bytes) -> bytes:
    return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def _pb_int(field: int, v: int) -> bytes:
    return _put_varint((field << 3) | 0) + _put_varint(v)


def build_object_graph(var_names: List[str], slot_names, slot: str = "momentum",
                       hyper: Tuple[str, ...] = OPTIMIZER_HYPER) -> bytes:
    """Serialized TrackableObjectGraph for model variables `var_names` (object paths without the VARIABLE_VALUE suffix),
    an `optimizer` child of the root with its hyper-parameter variables and one `slot` variable per entry of
    `slot_names` (a list of variable names for the one slot kind `slot`, or {slot kind: [variable names]} -- Adam has
    two, "m" and "v").  Node 0 is the root; nodes are numbered in breadth-first order."""
    slot_map = slot_names if isinstance(slot_names, dict) else ({slot: list(slot_names)} if slot_names else {})
    children: List[Dict[str, int]] = [{}]          # node -> {local name: node id}
    attr_key: Dict[int, str] = {}                  # variable node -> checkpoint key
    full_name: Dict[int, str] = {}

    def node_of(path: str, create=True) -> int:
        cur = 0
        for seg in path.split("/"):
            nxt = children[cur].get(seg)
            if nxt is None:
                if not create:
                    raise KeyError(path)
                children.append({})
                nxt = len(children) - 1
                children[cur][seg] = nxt
            cur = nxt
        return cur

    for name in var_names:
        nid = node_of(name)
        attr_key[nid] = name + SUFFIX
        full_name[nid] = name
    slots: List[Tuple[int, int, str]] = []
    if hyper or slot_map:
        for h in hyper:
            nid = node_of("optimizer/" + h)
            attr_key[nid] = f"optimizer/{h}{SUFFIX}"
            full_name[nid] = h
        opt = node_of("optimizer")
        for kind, names in slot_map.items():
            for name in names:        # slot variables hang off the optimizer node by reference, not by a named edge
                children.append({})
                sid = len(children) - 1
                attr_key[sid] = f"{name}/.OPTIMIZER_SLOT/optimizer/{kind}{SUFFIX}"
                full_name[sid] = f"{name}/{kind}"
                slots.append((node_of(name, create=False), sid, kind))
    else:
        opt = -1
    # breadth-first renumbering from the root (the order TF writes; restore does not depend on it)
    order, seen = [0], {0}
    i = 0
    while i < len(order):
        for _, c in sorted(children[order[i]].items()):
            if c not in seen:
                seen.add(c)
                order.append(c)
        i += 1
    order += [sid for _, sid, _ in slots]
    new_id = {old: new for new, old in enumerate(order)}
    out = bytearray()
    for old in order:
        node = bytearray()
        for local, c in sorted(children[old].items()):
            node += _pb_len(1, _pb_int(1, new_id[c]) + _pb_len(2, local.encode()))
        if old in attr_key:
            node += _pb_len(2, _pb_len(1, b"VARIABLE_VALUE") + _pb_len(2, full_name[old].encode()) +
                            _pb_len(3, attr_key[old].encode()))
        if old == opt:
            for var, sid, kind in slots:
                node += _pb_len(3, _pb_int(1, new_id[var]) + _pb_len(2, kind.encode()) + _pb_int(3, new_id[sid]))
        out += _pb_len(1, bytes(node))
    return bytes(out)


def parse_object_graph(buf: bytes) -> List[dict]:
    """TrackableObjectGraph bytes -> [{children: {local name: node}, attributes: {name: checkpoint key}, slots: [...]}]"""
    nodes = []
    for raw in _parse_proto(buf).get(1, []):
        f = _parse_proto(raw)
        ch = {}
        for c in f.get(1, []):
            cf = _parse_proto(c)
            ch[cf[2][0].decode()] = cf.get(1, [0])[0]
        at = {}
        for a_ in f.get(2, []):
            af = _parse_proto(a_)
            at[af[1][0].decode()] = af[3][0].decode()
        sl = []
        for s_ in f.get(3, []):
            sf = _parse_proto(s_)
            sl.append((sf.get(1, [0])[0], sf[2][0].decode(), sf.get(3, [0])[0]))
        nodes.append(dict(children=ch, attributes=at, slots=sl))
    return nodes


def _string_tensor_bytes(value: bytes) -> Tuple[bytes, int]:
    """Scalar DT_STRING in a tensor bundle (tensor_bundle.cc WriteStringTensor): varint length | masked CRC32C of the
    length (as uint32) | bytes; the entry checksum runs over length word, length checksum and bytes.  [TF-3p]"""
    ln = struct.pack("<I", len(value))
    c = crc32c(ln)
    cks = struct.pack("<I", mask_crc(c))
    c = crc32c(cks, c)
    c = crc32c(value, c)
    return _put_varint(len(value)) + cks + value, mask_crc(c)


def read_object_graph(path: str) -> List[dict]:
    """Parse the `_CHECKPOINTABLE_OBJECT_GRAPH` entry of a bundle (needs the data shard)."""
    prefix = resolve_prefix(path)
    header, entries = read_index(prefix + ".index")
    e = entries[OBJECT_GRAPH_KEY]
    fn = f"{prefix}.data-{e.shard_id:05d}-of-{header.get('num_shards', 1):05d}"
    with open(fn, "rb") as f:
        f.seek(e.offset)
        raw = f.read(e.size)
    ln, pos = _get_varint(raw, 0)
    body = raw[pos + 4:pos + 4 + ln]
    want = _string_tensor_bytes(body)
    if want[0] != raw or want[1] != e.crc32c:
        raise ValueError(f"{OBJECT_GRAPH_KEY}: string-tensor framing / checksum mismatch")
    return parse_object_graph(body)


def write_checkpoint(prefix: str, state: Dict[str, torch.Tensor], specs: Dict[str, object],
                     momentum: Optional[Dict[str, torch.Tensor]] = None, optimizer_hyper: Optional[Dict[str, float]] = None,
                     slots: Optional[Dict[str, Dict[str, torch.Tensor]]] = None):
    """Write ``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` + the ``checkpoint`` state file with the
    reference's keys and TF layouts, including the ``_CHECKPOINTABLE_OBJECT_GRAPH`` entry Keras' object-based
    ``load_weights`` restores by (build_object_graph) and the optimizer's hyper-parameter variables
    (``optimizer/{iter,learning_rate,momentum,decay}``: `optimizer_hyper`, defaults 0 / 0.0).  `slots` = {slot kind:
    {variable: tensor}} writes other slot variables than SGD's `momentum`: with the kinds "m" / "v" the optimizer is
    written as Keras' Adam (hyper variables ADAM_HYPER)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = []
    slots = dict(slots or {})
    if momentum:
        slots["momentum"] = momentum
    slot_names = {kind: [] for kind in slots}
    for name, spec in specs.items():
        items.append((name + SUFFIX, to_tf(spec, state[name].detach().float().cpu())))
        for kind, tensors in slots.items():
            if name in tensors:
                items.append((f"{name}/.OPTIMIZER_SLOT/optimizer/{kind}{SUFFIX}",
                              to_tf(spec, tensors[name].detach().float().cpu())))
                slot_names[kind].append(name)
    slot_names = {k: v for k, v in slot_names.items() if v}
    adam = "m" in slots or "v" in slots
    hyper_names = ADAM_HYPER if adam else OPTIMIZER_HYPER
    hyper = dict(decay=0.0, iter=0, learning_rate=0.0, beta_1=0.9, beta_2=0.999) if adam else \
        dict(decay=0.0, iter=0, learning_rate=0.0, momentum=0.0)
    hyper.update({k: v for k, v in (optimizer_hyper or {}).items() if k in hyper_names})
    raw_items = []     # (key, dtype, shape, raw bytes, masked crc)
    for key, t in items:
        raw = t.numpy().astype("<f4").tobytes()
        raw_items.append((key, DT_FLOAT, tuple(t.shape), raw, mask_crc(crc32c(raw))))
    for h in hyper_names:
        raw = struct.pack("<q", int(hyper[h])) if h == "iter" else struct.pack("<f", float(hyper[h]))
        raw_items.append((f"optimizer/{h}{SUFFIX}", DT_INT64 if h == "iter" else DT_FLOAT, (), raw, mask_crc(crc32c(raw))))
    graph = build_object_graph(list(specs), slot_names, hyper=hyper_names)
    graw, gcrc = _string_tensor_bytes(graph)
    raw_items.append((OBJECT_GRAPH_KEY, DT_STRING, (), graw, gcrc))
    raw_items.sort(key=lambda it: it[0])
    entries = {}
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for key, dt, shape, raw, crc in raw_items:
            f.write(raw)
            entries[key] = BundleEntry(dt, shape, 0, off, len(raw), crc)
            off += len(raw)
    write_index(prefix + ".index", entries)
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')
    return prefix


def _flat_slot(model, flat, k):
    o = model._offsets[k]
    return flat[o:o + model.params[k].numel()].view(model.params[k].shape)


def load_tf_checkpoint(model, path, expect_partial=True, optimizer=None):
    """Variables + optimizer state (reference train.py:131-136 `model.load_weights(latest)`): SGD `momentum` slots or
    Adam `m` / `v` slots into the model's flat slot buffers; the optimizer's hyper-parameter variables (`iter`, ...) are
    left in `model.optimizer_state` for the trainer (Trainer.resume restores its step counter from `iter`).

    optimizer: the optimizer branch that will USE the slots ("sgd" | "adam" | None = whatever the bundle holds).  Slots of
    the other branch are not installed (Keras restores the variables and leaves a new optimizer's slots at zero): Adam's first
    moment is never SGD momentum or vice versa.  Whatever is installed is recorded in `model.slot_kind`, which
    `apply_sgd` / `apply_adam` check before their first use of the buffers."""
    sd, slots, hyper = read_checkpoint(path, model.specs, expect_partial=expect_partial, with_slots=("momentum", "m", "v"))
    model.load_state_dict(sd)
    kind = "adam" if slots["m"] or slots["v"] else "sgd" if slots["momentum"] else None
    model.flat_velocity.zero_()
    if getattr(model, "flat_second", None) is not None:
        model.flat_second.zero_()
    install = kind is not None and optimizer in (None, kind)
    if install:
        for k, v in (slots["m"] or slots["momentum"]).items():
            if k in model.grads:
                _flat_slot(model, model.flat_velocity, k).copy_(v)
        if slots["v"]:
            if getattr(model, "flat_second", None) is None:
                model.flat_second = torch.zeros_like(model.flat_velocity)
            for k, v in slots["v"].items():
                if k in model.grads:
                    _flat_slot(model, model.flat_second, k).copy_(v)
    model.slot_kind = kind if install else None
    model.optimizer_state = dict(hyper=hyper, kind=kind)
    return model


def save_tf_checkpoint(model, prefix, optimizer_hyper=None, optimizer="sgd"):
    """optimizer="sgd": the first slot buffer is written as Keras SGD's `momentum` slots (the released bundles' layout);
    "adam": both moments as Keras Adam's `m` / `v` slots with Adam's hyper variables."""
    first = {k: _flat_slot(model, model.flat_velocity, k) for k in model.grads}
    if optimizer == "adam":
        second = getattr(model, "flat_second", None)
        if second is None:
            second = torch.zeros_like(model.flat_velocity)
        return write_checkpoint(prefix, model.state_dict(), model.specs, optimizer_hyper=optimizer_hyper,
                                slots={"m": first, "v": {k: _flat_slot(model, second, k) for k in model.grads}})
    return write_checkpoint(prefix, model.state_dict(), model.specs, momentum=first, optimizer_hyper=optimizer_hyper)
