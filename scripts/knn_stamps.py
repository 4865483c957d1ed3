"""Development (GPU box): where the fused feature-space kNN kernel's time goes, per workgroup and phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _dev import use_dev_lib
lib = use_dev_lib()
from tgpose_amd import ops
dev = "cuda:0"
B, n, d, k = 32, 1028, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 20
x = torch.relu(torch.randn(B, n, d, device=dev) * 0.7 + 0.2)
for _ in range(3):
    ops.knn_feat(x, k)
st = torch.zeros(4096, 12, dtype=torch.int64, device=dev)
lib.tgp_debug_set_knn_stamps.argtypes = [ctypes.c_void_p]
lib.tgp_debug_set_knn_stamps(ctypes.c_void_p(st.data_ptr()))
ops.knn_feat(x, k)
torch.cuda.synchronize()
lib.tgp_debug_set_knn_stamps(None)
s = st.cpu().numpy().astype(np.float64) / 100.0          # us
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
print("workgroups %d  kernel span %.1f us" % (len(s), s.max() - t0))
print("distance phase (wave 0) %.1f us   wait at barrier %.1f   selection (wave 0's four rows) %.1f   (medians)" % (
    np.median(s[:, 1] - s[:, 0]), np.median(s[:, 2] - s[:, 1]), np.median(s[:, 3] - s[:, 2])))
print("start times, every 128th:", np.sort(s[:, 0] - t0)[::128])
print("phase-1 end of waves 0..7 after the workgroup's start (medians, us):", np.round(np.median(s[:, 4:12] - s[:, :1], axis=0), 1))
