"""Development (GPU box): where the fused heads kernel's time goes -- wave 0's accumulated time per phase, per workgroup."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _dev import use_dev_lib
lib = use_dev_lib()
from bench import synth_batch, N_POINTS
from tgpose_amd import PoseNet9D, seeded_state_dict, FLAGS, engine
dev = "cuda:0"
net = PoseNet9D(); net.load_state_dict(seeded_state_dict(0)); net = net.to(dev).eval()
FLAGS.train = 0
engine.BRANCH_STREAMS = False
pts, obj = synth_batch(32, N_POINTS, 100)
pts, obj = pts.to(dev), obj.to(dev)
for _ in range(3):
    net(pts, obj)
st = torch.zeros(1024, 6, dtype=torch.int64, device=dev)
lib.tgp_debug_set_heads_stamps.argtypes = [ctypes.c_void_p]
lib.tgp_debug_set_heads_stamps(ctypes.c_void_p(st.data_ptr()))
net(pts, obj)
torch.cuda.synchronize()
lib.tgp_debug_set_heads_stamps(None)
s = st.cpu().numpy().astype(np.float64) / 100.0          # us (100 MHz counter)
s = s[s[:, 5] > 0]
names = ["phase 1 (conv1 MFMAs)", "epilogue 1", "DMA issue", "phase 2 (conv2 MFMAs)", "barrier"]
print("workgroups %d; per workgroup, wave 0, summed over the 32 channel blocks (median us):" % len(s))
for i, n in enumerate(names):
    print("  %-24s %7.1f" % (n, np.median(s[:, i])))
print("  total                    %7.1f   kernel span %.1f us" % (np.median(s[:, :5].sum(1)), s[:, 5].max() - s[:, 5].min() + np.median(s[:, :5].sum(1))))
