"""Development (GPU box): per-launch table of every GEMM of one eval forward at B=32, N=1028 -- shape, microseconds, TF-equivalent."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch, N_POINTS
from tgpose_amd import PoseNet9D, seeded_state_dict, FLAGS, ops, engine

if os.environ.get("TGP_SPLIT_VARIANT"):       # development A/B of split-GEMM variants (csrc/gemm.hip)
    import ctypes
    from tgpose_amd import _lib
    from _dev import use_dev_lib
    use_dev_lib().tgp_debug_set_split_variant(int(os.environ["TGP_SPLIT_VARIANT"]))
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = PoseNet9D()
net.load_state_dict(seeded_state_dict(0), strict=True)
net = net.to(dev).eval()
FLAGS.train = 0
engine.BRANCH_STREAMS = False
pts, obj = synth_batch(B, N_POINTS, 100)
pts, obj = pts.to(dev), obj.to(dev)
for _ in range(3):
    net(pts, obj)
torch.cuda.synchronize()
ops.GEMM_TIMER, ops.GEMM_TIMER_ALL = [], True
R = 10
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(R):
    net(pts, obj)
t1.record()
torch.cuda.synchronize()
timer, ops.GEMM_TIMER = ops.GEMM_TIMER, None
per = len(timer) // R
acc = collections.OrderedDict()
for i, (e0, e1, fl, shape, *_) in enumerate(timer):
    k = (i % per, shape)
    acc.setdefault(k, [0.0, fl])[0] += e0.elapsed_time(e1) * 1e3 / R
tot = 0.0
print("forward %.3f ms (eager, serial)" % (t0.elapsed_time(t1) / R))
print("%3s %7s %6s %6s %3s %9s %8s" % ("#", "M", "N", "K", "b", "us", "TF-eq"))
for (i, (M, N, K, b)), (us, fl) in acc.items():
    tot += us
    print("%3d %7d %6d %6d %3d %9.1f %8.1f" % (i, M, N, K, b, us, fl / us * 1e-6))
print("sum of GEMM launches (event-bracketed, includes launch gaps): %.1f us" % tot)
