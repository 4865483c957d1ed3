"""Development (GPU box): node census of the captured trainer step (kernel nodes per replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import train_batch, N_POINTS
from tgpose_amd.trainer.RL_TDA import RT_TDA_Trainer
from tgpose_amd import seeded_state_dict, FLAGS

dev = torch.device("cuda:0")
tr = RT_TDA_Trainer(device=dev)
tr.init_network('RL_TDA')
tr.init_loss()
tr.net1.load_state_dict(seeded_state_dict(0))
tr.net2.load_state_dict(seeded_state_dict(1, only_encoder=True))
tr.net1.train(), tr.net2.train()
tr.optimizer = torch.optim.SGD(tr.net1.parameters(), lr=1e-5, momentum=0.9)
db = {k: v.to(dev) for k, v in train_batch(32, N_POINTS, 7).items()}
for overlap in (False, True):
    step = tr.graphed_step(db, overlap=overlap)
    g = step.graph
    print("overlap=%s: nodes (kernels, memcpys, memsets, other) = %s%s" % (overlap, g.nodes, (" + second segment %s" % (g.nodes2,)) if getattr(g, "nodes2", None) else ""))
    loss = step()
    tr.finish_step(total=loss)
    torch.cuda.synchronize()
    del step, g
FLAGS.train = 0
