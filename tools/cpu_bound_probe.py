import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import x3d_tf_amd as x
from x3d_tf_amd.model import X3D
for variant, views, t, s, dt in [("XS", 10, 4, 160, torch.float32), ("XS", 10, 4, 160, torch.float16), ("S", 30, 13, 182, torch.float16), ("M", 30, 16, 256, torch.float16)]:
    cfg = x.get_config(variant, ["TEST.NUM_TEMPORAL_VIEWS", views, "TEST.NUM_SPATIAL_CROPS", 1])
    m = X3D(cfg, dtype=dt, device="cuda:0")
    clips = torch.randn(views, t, s, s, 3, device="cuda").to(dt)
    for _ in range(3): m(clips, training=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = m(clips, training=False)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
    # GPU time via events around the whole forward
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pl = m._plan(views, t, s, s, False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pl.run(pl.fwd)
    cpu_issue = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    print(f"{variant} {views} views {t}x{s} {dt}: wall {wall*1e3:.2f} ms per video; host time to issue the {len(pl.fwd)} launches {cpu_issue*1e3:.2f} ms", flush=True)
    del m
