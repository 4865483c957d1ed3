"""Development (GPU box): which torch-side launches of the trainer's step cost time -- kernel time of ATen ops by op and shape."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import train_batch, N_POINTS
from tgpose_amd.trainer.RL_TDA import RT_TDA_Trainer, total_loss
from tgpose_amd import seeded_state_dict, FLAGS

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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_iteration(db)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("ATen ops with device time: %.2f ms in one eager step" % (tot / 1e3))
for t, c, k, s in rows[:70]:
    print("%8.1f us  x%-4d %-28s %s" % (t, c, k, s))
