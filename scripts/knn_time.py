"""Development (GPU box): feature-space kNN timings at the forward's shapes (B = 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import ops

dev = "cuda:0"
for (B, n, d, k) in ((32, 1028, 128, 20), (32, 257, 128, 20), (32, 257, 256, 20), (32, 64, 256, 8), (256, 1028, 128, 20)):
    x = torch.relu(torch.randn(B, n, d, device=dev) * 0.7 + 0.2)
    ref = None
    for form in (2, 3, 0):          # 16-row blocks, two workgroups per CU | producer / consumer waves | the library's choice
        for _ in range(3):
            idx = ops.knn_feat(x, k, form=form)
        torch.cuda.synchronize()
        ref = idx.clone() if ref is None else ref
        assert torch.equal(idx, ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            idx = ops.knn_feat(x, k, form=form)
        e1.record()
        torch.cuda.synchronize()
        print("knn_feat B=%d n=%d d=%d k=%d form %d: %.1f us per call (prep + fused distance/selection)" % (B, n, d, k, form, e0.elapsed_time(e1) * 1e3 / 20), flush=True)
