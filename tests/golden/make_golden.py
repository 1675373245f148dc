"""Regenerates the fixtures under tests/golden/.  Run in the BUILD container only (it reads the reference
checkout at /root/reference, which does not exist on the GPU box); the outputs are committed.

  summaries.json             rows + totals of reference models/X3D-*/X3D_*.txt (Keras `summary()` dumps)
  model_index_manifest.json  key / dtype / shape / offset / size of every entry of reference
                             models/X3D-M/model.index (+ check that XS and S have identical layouts)
  oracle_xs_forward.json     one whole-model X3D-XS inference (10 views of 4x160x160, seeded synthetic weights
                             and input) through oracle/x3d_oracle.py: logits and probabilities
  oracle_train_tiny.json     loss / selected gradient norms of one oracle training step on a tiny clip batch
"""
import json
import os
import re
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def summaries():
    out = {}
    for v in ["XS", "S", "M", "L", "XL"]:
        rows, totals = [], {}
        for line in open(f"{REF}/models/X3D-{v}/X3D_{v}.txt"):
            m = re.match(r"^(\S+) \((\w+)\)\s+(\[?\(None.*?\)\]?)\s+(\d+)\s*$", line)
            if m:
                shape = [int(s) for s in re.findall(r"\d+", m.group(3))]
                rows.append(dict(name=m.group(1), kind=m.group(2), shape=shape, params=int(m.group(4))))
            m = re.match(r"^(Total|Trainable|Non-trainable) params: ([\d,]+)", line)
            if m:
                totals[m.group(1).lower()] = int(m.group(2).replace(",", ""))
        out[v] = dict(rows=rows, totals=totals)
    return out


def manifest():
    from x3d_tf_amd import checkpoint as ck
    res = {}
    for v in ["XS", "S", "M"]:
        _, ent = ck.read_index(f"{REF}/models/X3D-{v}/model.index")
        res[v] = ent
    lay = lambda e: (e.dtype, tuple(e.shape), e.offset, e.size)
    same = all(set(res[v]) == set(res["M"]) and all(lay(res[v][k]) == lay(res["M"][k]) for k in res["M"])
               for v in ["XS", "S"])
    entries = [dict(key=k, dtype=e.dtype, shape=list(e.shape), offset=e.offset, size=e.size)
               for k, e in sorted(res["M"].items())]
    return dict(source="models/X3D-M/model.index", identical_layout_XS_S_M=same, entries=entries)


def oracle_vectors():
    import x3d_tf_amd as x
    from x3d_tf_amd.params import init_params, randomize_bn_
    from oracle import x3d_oracle as O
    cfg = x.get_config("XS")
    arch = x.build_arch(cfg)
    p = randomize_bn_(init_params(arch, seed=3), seed=4)
    torch.manual_seed(0)
    xin = torch.randn(10, 4, 160, 160, 3)
    probs, logits = O.forward(p, xin, arch, training=False, return_logits=True)
    fwd = dict(config="XS", param_seed=3, bn_seed=4, input_seed=0, input_shape=[10, 4, 160, 160, 3],
               logits_view0=logits[0].tolist(), probs=probs[0].tolist())
    torch.manual_seed(1)
    xt = torch.randn(4, 4, 64, 64, 3)
    labels = torch.randint(0, 400, (4,))
    mask = (torch.rand(4, 2048) >= 0.5).float()
    r = O.train_step({k: v.clone() for k, v in p.items()}, xt, labels, arch, lr=None, dropout_mask=mask,
                     apply_update=False)
    keys = ["conv1/conv_s/kernel", "conv1/conv_t/kernel", "conv1/bn/gamma",
            "stages/0/stage/layer_with_weights-0/bottleneck/a/kernel",
            "stages/0/stage/layer_with_weights-0/bottleneck/b/kernel",
            "stages/0/stage/layer_with_weights-0/bottleneck/se_fc1/kernel",
            "stages/0/stage/layer_with_weights-0/residual/kernel",
            "stages/2/stage/layer_with_weights-3/bottleneck/c/kernel",
            "stages/3/stage/layer_with_weights-6/bottleneck/bn_c/beta", "fc1/kernel", "fc2/bias"]
    train = dict(config="XS", param_seed=3, bn_seed=4, input_seed=1, input_shape=[4, 4, 64, 64, 3],
                 labels=labels.tolist(), loss=r["loss"].item(), ce=r["ce"].item(), reg=r["reg"].item(),
                 grad_l2={k: r["grads"][k].double().norm().item() for k in keys},
                 grad_first={k: r["grads"][k].reshape(-1)[:4].tolist() for k in keys})
    return fwd, train


if __name__ == "__main__":
    json.dump(summaries(), open(f"{HERE}/summaries.json", "w"), indent=1)
    json.dump(manifest(), open(f"{HERE}/model_index_manifest.json", "w"))
    fwd, train = oracle_vectors()
    json.dump(fwd, open(f"{HERE}/oracle_xs_forward.json", "w"))
    json.dump(train, open(f"{HERE}/oracle_train_tiny.json", "w"), indent=1)
    print("ok")
