"""(argument "wgrad" first: the weight-gradient kernel, build with "-DWGP_STAMPS" pw_wgrad.hip)
Phase timeline of ONE workgroup of the fp32 pipelined pointwise kernel (pw_gemm_f32p.h) from in-kernel s_memtime stamps.
Build host:  tools/build_variant.sh f32pstamp "-DF32P_EXP=256" pw_fwd.hip
GPU box:     X3D_HIP_LIB=$PWD/x3d-tf_amd/libx3d_hip_f32pstamp.so python tools/f32p_stamps.py [cin cout t h w [swish]]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x3d_tf_amd import ops, hip
dev = torch.device("cuda:0")
wgrad = sys.argv[1:2] == ["wgrad"]
if wgrad:
    del sys.argv[1]
cin, cout, t, h, w = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (96, 216, 13, 10, 10)
swish = len(sys.argv) > 6
N = 32
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((N, cin, t, h, w), generator=g, device=dev)
wt = torch.randn((cout, cin), generator=g, device=dev) * 0.1
y = torch.empty((N, cout, t, h, w), device=dev)
st = ops.stats_buffer(cout, dev)
ss = torch.randn((cin, 2), generator=g, device=dev)
gate = torch.rand((N, cin), generator=g, device=dev)
fn = (lambda: ops.pw_fwd(x, wt, y=y, stats=st, in_ss=ss, in_gate=gate, in_act=2)) if swish else (lambda: ops.pw_fwd(x, wt, y=y, stats=st))
if wgrad:
    gy = torch.randn((N, cout, t, h, w), generator=g, device=dev); yraw = torch.randn((N, cout, t, h, w), generator=g, device=dev)
    coef = torch.randn((cout, 4), generator=g, device=dev); dw = torch.zeros((cout, cin), device=dev)
    fn = (lambda: ops.pw_wgrad(gy, yraw, coef, x, dw, in_ss=ss, in_gate=gate, in_act=2)) if swish else (lambda: ops.pw_wgrad(gy, yraw, coef, x, dw))
for _ in range(5):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
lib = ctypes.CDLL(hip.LIB_PATH)
buf = (ctypes.c_ulonglong * 128)()
assert (lib.x3d_debug_wgp_stamps if wgrad else lib.x3d_debug_f32p_stamps)(buf) == 0
for wg in range(1 if wgrad else 2):
    s = list(buf[wg * 64:(wg + 1) * 64])
    t0 = s[0]
    names = {0: "start", 1: "weights + tables + first issues", 2: "barrier", 3: "first commit", 60: "loop end", 61: "kernel end (flush)"}
    print(f"workgroup {'0' if wg == 0 else 'last'}: event-timed launch {e0.elapsed_time(e1) * 1e3:.1f} us")
    prev = t0
    for i, v in enumerate(s):
        if v == 0 or v < t0:
            continue
        print(f"  [{i:2d}] {names.get(i, 'loop round ' + str((i - 4))):34s} +{v - prev:7d}   t = {v - t0:7d} ticks")
        prev = v
