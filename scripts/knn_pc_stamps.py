"""Development library: where the producer / consumer form of the feature-space kNN spends its cycles (per wave: inside its role's body, inside the barrier)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _dev import use_dev_lib
dev_lib = use_dev_lib()
import torch
from tgpose_amd import ops
dev = "cuda:0"
B, n, d, k = 32, 1028, 128, 20
x = torch.relu(torch.randn(B, n, d, device=dev) * 0.7 + 0.2)
for _ in range(3):
    ops.knn_feat(x, k, form=3)
st = torch.zeros(B * 8 * 24, device=dev, dtype=torch.int64)
dev_lib.tgp_debug_set_knn_stamps(ctypes.c_void_p(st.data_ptr()))
ops.knn_feat(x, k, form=3)
torch.cuda.synchronize()
dev_lib.tgp_debug_set_knn_stamps(None)
t = st[: B * 8 * 16].view(-1, 8, 2).cpu().double()
phs = st[B * 8 * 16:].view(-1, 8).cpu().double()
for w in range(8):
    print("wave %d (%s): body %8.0f cycles  barrier %8.0f   (median over %d workgroups)" % (w, "producer" if w < 4 else "consumer", t[:, w, 0].median(), t[:, w, 1].median(), t.shape[0]))
print("consumer wave 5, cycles over its 32 rows: keys %.0f | lane minima, ranks, bound %.0f | count + prefix %.0f | compaction %.0f | survivor ranks + store %.0f"
      % tuple(phs[:, i].median() for i in range(5)))
