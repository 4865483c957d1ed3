"""Development: eager single backward vs eager two-call backward (EncoderCut): per-parameter relative difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_parity import _trainer, _step_db, g
from tgpose_amd.autograd import EncoderCut
from tgpose_amd.trainer.RL_TDA import total_loss

B, N = 6, 512
db = {k: g(v) for k, v in _step_db([0, 1, 2, 3, 4, 5], N, 37).items()}
torch.manual_seed(19)
pair = []
for _ in range(2):
    i1 = torch.randperm(N)[: N // 4]
    pair.append((i1, torch.randperm(i1.numel())[: i1.numel() // 4]))
res = {}
for mode in ("single", "single2", "cut"):
    tr = _trainer(17)
    cut = EncoderCut() if mode == "cut" else None
    _, ld = tr.RL_TDA_train_step(db, sample_idx=pair, cut=cut)
    t = total_loss(ld)
    t.backward()
    if cut is not None:
        print("leaf grad norm", cut.leaf.grad.norm().item(), "feat requires_grad", cut.feat.requires_grad)
        cut.backward_encoder()
    res[mode] = (t.item(), {k: p.grad.clone() for k, p in tr.net1.named_parameters() if p.grad is not None})
for a, b in (("single", "single2"), ("single", "cut")):
    print(a, "vs", b, res[a][0], res[b][0])
    worst = sorted(((res[b][1][k] - v).abs().max().item() / (v.abs().max().item() + 1e-12), k) for k, v in res[a][1].items() if k in res[b][1])[-6:]
    for w in worst:
        print("   %.2e  %s" % w)
    print("   missing in", b, [k for k in res[a][1] if k not in res[b][1]][:5])
