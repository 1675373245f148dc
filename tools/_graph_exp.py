"""Experiment: one train step captured in a HIP graph (torch.cuda.CUDAGraph) vs eager replay of the plan."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x3d_tf_amd as x3d
from x3d_tf_amd.train import Trainer

dev = torch.device("cuda:0")
cfg = x3d.get_config("M")
m = x3d.X3D(cfg, dtype=torch.bfloat16, device=dev)
tr = Trainer(m, cfg)
B = 64
clips = torch.randn(B, 16, 224, 224, 3, device=dev).bfloat16()
labels = torch.randint(0, 400, (B,), device=dev)
for _ in range(3):
    tr.step(clips, labels, 0.01)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.step(clips, labels, 0.01)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) * 100)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    tr.step(clips, labels, 0.01)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        tr.step(clips, labels, 0.01)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) * 100)
