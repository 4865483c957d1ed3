"""Times the ORL pooling launch (tgp_orl_rowbias_fused) on the forward's five shapes, neighbour lists staged in LDS against read from memory
(development library), and checks both forms leave the same bits.   python scripts/orl_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _dev import use_dev_lib  # noqa: E402

dev_lib = use_dev_lib()
from tgpose_amd import ops  # noqa: E402

dev = "cuda:0"
gen = torch.Generator().manual_seed(0)
for B, n, k, C in ((32, 1028, 20, 128), (32, 257, 20, 256), (32, 64, 8, 512)):
    feat = torch.randn(B, n, C, generator=gen).to(dev)
    idx = torch.stack([torch.stack([torch.randperm(n, generator=gen)[:k] for _ in range(n)]) for _ in range(B)]).int().to(dev)
    w2t = (torch.randn(C, C, generator=gen) / C ** 0.5).to(dev)
    res = {}
    for mode in (0, 1):
        dev_lib.tgp_debug_set_orl_idx(mode)
        pl = ops.Planes(B * n, C, dev)
        tickets = torch.zeros(B, device=dev, dtype=torch.int32)
        out = ops.orl_rowbias(feat, idx, w2t, planes=pl, tickets=tickets)
        torch.cuda.synchronize()
        res[mode] = (out[0].clone() if isinstance(out, (tuple, list)) else out.clone(), pl.buf.clone())
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.orl_rowbias(feat, idx, w2t, planes=pl, tickets=tickets)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print("B=%d n=%d k=%d C=%d  lists %s: %6.1f us" % (B, n, k, C, "in LDS   " if mode else "in memory", min(ts)), flush=True)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
print("both forms bit-identical")
