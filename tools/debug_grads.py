"""Debug helper: per-parameter gradient error of the HIP path vs the oracle, in backward order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import x3d_tf_amd as x
from x3d_tf_amd.params import init_params, randomize_bn_
from x3d_tf_amd.model import X3D
from oracle import x3d_oracle as O

name, n, t, s = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = torch.bfloat16 if len(sys.argv) > 5 and sys.argv[5] == "bf16" else torch.float32
cfg = x.get_config(name); arch = x.build_arch(cfg)
params = randomize_bn_(init_params(arch, seed=3), seed=4)
if len(sys.argv) > 6:
    for k in params:
        if k.endswith("bn_c/gamma"): params[k] *= float(sys.argv[6])
torch.manual_seed(1)
xin = torch.randn(n, t, s, s, 3)
if dtype == torch.bfloat16: xin = xin.bfloat16().float()
labels = torch.randint(0, arch.num_classes, (n,))
mask = (torch.rand(n, arch.fc1_out) >= arch.dropout_rate).float()
m = X3D(cfg, dtype=dtype, device="cuda:0"); m.load_state_dict(params); m.set_dropout_mask(mask)
pl = m.forward_backward(xin.cuda(), labels.cuda()); torch.cuda.synchronize()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import hip_relu_masks
r = O.train_step({k: v.clone() for k, v in params.items()}, xin, labels, arch, lr=None, dropout_mask=mask, apply_update=False,
                 relu_masks=hip_relu_masks(pl), storage=dtype)
print("probs err", (pl.probs.cpu() - r["probs"]).abs().max().item())
names = list(r["grads"].keys())[::-1]
for k in names:
    g = m.grads[k].cpu().double()
    if m.specs[k].l2: g = g + 2 * arch.weight_decay * params[k].double()
    gr = r["grads"][k].double()
    sc = gr.abs().max().item() + 1e-12
    err = (g - gr).abs().max().item() / sc
    l2 = ((g - gr).norm() / (gr.norm() + 1e-30)).item()
    flag = " <<<" if err > 2e-3 else ""
    print(f"{err:9.2e} {l2:9.2e} {sc:9.2e} {k}{flag}")

# ReLU-mask agreement per block (a ReLU flip on an element with |z| ~ 1e-7 changes the gradient discontinuously)
taps = {}
st = O.BNState()
O.forward({k: v.clone() for k, v in params.items()}, xin, arch, training=True, dropout_mask=mask, state=st, taps=taps)
for B in pl.blocks:
    pre = O.block_prefix(B.spec)
    a_ref = taps[pre + "/a_raw"]
    mean, var = st.batch_stats[pre + "/bottleneck/bn_a"]
    g_, b_ = params[pre + "/bottleneck/bn_a/gamma"], params[pre + "/bottleneck/bn_a/beta"]
    z_ref = (a_ref - mean.view(1, -1, 1, 1, 1)) * (torch.rsqrt(var + arch.bn_eps) * g_).view(1, -1, 1, 1, 1) + b_.view(1, -1, 1, 1, 1)
    ss = B.bn_a.ss.cpu()
    z_hip = B.a_raw.float().cpu() * ss[:, 0].view(1, -1, 1, 1, 1) + ss[:, 1].view(1, -1, 1, 1, 1)
    flips = ((z_ref > 0) != (z_hip > 0)).sum().item()
    yflip = ((taps[pre + "/out"] > 0) != (B.y.float().cpu() > 0)).sum().item()
    print(f"{pre}: relu_a flips {flips} (min|z_ref| {z_ref.abs().min().item():.2e}), out flips {yflip}")
