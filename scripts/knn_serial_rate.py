"""Development library: how many rows of the feature-space kNN take the serial selection (more than 64 survivors under the rank-counting
bound) -- on the benchmark's eval forward (B = 32, N = 1028, seeded weights) and on the synthetic features of scripts/knn_time.py."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _dev import use_dev_lib
dev_lib = use_dev_lib()
import torch
from tgpose_amd import ops, PoseNet9D, seeded_state_dict, FLAGS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import synth_points
dev = "cuda:0"
cnt = ctypes.c_ulonglong(0)
def read(reset=True):
    assert dev_lib.tgp_debug_knn_serial_rows(ctypes.byref(cnt), int(reset)) == 0
    return cnt.value
net = PoseNet9D(); net.load_state_dict(seeded_state_dict(0), strict=True); net = net.to(dev).eval(); FLAGS.train = 0
pts, obj = synth_points(32, 1028, 100)
with torch.no_grad():
    net(pts.to(dev), obj.to(dev))
torch.cuda.synchronize(); read()
with torch.no_grad():
    net(pts.to(dev), obj.to(dev))
torch.cuda.synchronize()
rows = 32 * (1028 + 257 + 257 + 64)
print("eval forward, B = 32, N = 1028: %d of %d feature-kNN rows took the serial selection (%.3f %%)" % (read(), rows, 100.0 * cnt.value / rows))
x = torch.relu(torch.randn(32, 1028, 128, device=dev) * 0.7 + 0.2)
ops.knn_feat(x, 20); torch.cuda.synchronize(); read()
ops.knn_feat(x, 20); torch.cuda.synchronize()
print("relu(0.7 randn + 0.2) features (scripts/knn_time.py): %d of %d rows" % (read(), 32 * 1028))
