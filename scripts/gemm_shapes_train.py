"""Development (GPU box): per-launch table of every NT GEMM launch (ops.gemm) of one eager trainer step at B=32, N=1028."""
import os, sys, collections
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
ops.GEMM_TIMER, ops.GEMM_TIMER_ALL = [], True
tr.train_iteration(db)
torch.cuda.synchronize()
timer, ops.GEMM_TIMER = ops.GEMM_TIMER, None
tot = 0.0
print("%3s %7s %6s %6s %3s %9s %8s" % ("#", "M", "N", "K", "b", "us", "TF"))
for i, (e0, e1, fl, shape, *_) in enumerate(timer):
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print("%3d %7d %6d %6d %3d %9.1f %8.1f" % ((i,) + tuple(shape) + (us, fl / us * 1e-6)))
print("sum %.1f us over %d launches" % (tot, len(timer)))
