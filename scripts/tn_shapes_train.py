"""Development (GPU box): every dW = dy^T x of one eager trainer step -- shape, which path, microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import train_batch, N_POINTS
from tgpose_amd.trainer.RL_TDA import RT_TDA_Trainer
from tgpose_amd import seeded_state_dict, ops

dev = "cuda:0"
tr = RT_TDA_Trainer(device=dev)
tr.init_network('RL_TDA')
tr.init_loss()
tr.net1.load_state_dict(seeded_state_dict(0), strict=True)
tr.net2.load_state_dict(seeded_state_dict(1, only_encoder=True), strict=True)
tr.net1.train(), tr.net2.train()
tr.optimizer = torch.optim.SGD(tr.net1.parameters(), lr=1e-5, momentum=0.9)
db = {k: v.to(dev) for k, v in train_batch(32, N_POINTS, 1).items()}
for _ in range(2):
    tr.train_iteration(db)
torch.cuda.synchronize()
log = []
orig = ops.gemm_tn


def spy(a, b, out=None, accumulate=False, scale=None):
    rows, N, K = a.numel() // a.shape[-1], a.shape[-1], b.shape[-1]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(a, b, out=out, accumulate=accumulate, scale=scale)
    e1.record()
    log.append((rows, N, K, "split" if ops.tn_split_ok(rows, N, K) else "fp32", scale is not None, e0, e1))
    return r


ops.gemm_tn = spy
import tgpose_amd.autograd as A
tr.train_iteration(db)
torch.cuda.synchronize()
tot = {"split": 0.0, "fp32": 0.0}
for rows, N, K, path, hint, e0, e1 in log:
    us = e0.elapsed_time(e1) * 1e3
    tot[path] += us
    print("rows %6d N %5d K %5d  %-5s scale given %d  %7.1f us" % (rows, N, K, path, hint, us))
print(tot)
