"""Which piece of the loss bundle survives hipGraph capture?  python scripts/capture_probe.py <piece> (run each in its own process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import synth_loss_batch
from tgpose_amd.losses.TDA_loss_sym_recon import TDA_loss
from tgpose_amd.losses.consistency_loss import feat_consistency_loss, prop_sym_matching_loss

piece, bwd = sys.argv[1], int(sys.argv[2])
BIG = "big" in sys.argv[3:]
OPT = set(sys.argv[3:])
dev = "cuda:0"
pred, gt, sym, extra = synth_loss_batch(seed=5, B=300, N=1028, D=2500, C=1286) if BIG else synth_loss_batch(seed=5, B=40, N=256, D=100, C=128)
dp = {k: v.to(dev).requires_grad_(True) for k, v in pred.items()}
dg = {k: v.to(dev) for k, v in gt.items()}
dx = {k: v.to(dev) for k, v in extra.items()}
dsym = sym.to(dev)
mod = TDA_loss()
f1 = dx["feat1"].clone().requires_grad_(True)


def step():
    if piece == "pose":
        l = sum(v.sum() for v in mod(['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con'], dp, dg, dsym).values())
    elif piece == "ph":
        l = sum(v.sum() for v in mod(['TDA_h1', 'TDA_h2'], dp, dg, dsym).values())
    elif piece == "phcate":
        l = sum(v.sum() for v in mod(['TDA_h1_cate'], dp, dg, dsym).values())
    elif piece == "sym":
        l = prop_sym_matching_loss(dp["Recon"], dx["recon2"], dg["R"], dg["Tran"], dsym)
    elif piece == "feat":
        l = feat_consistency_loss(f1, dx["feat2"])
    elif piece == "all":
        res = mod(['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con', 'TDA_h1', 'TDA_h2', 'TDA_h1_cate',
                   'TDA_h2_cate', 'Prop_sym'], dp, dg, dsym)
        l = sum(v.sum() for v in res.values()) + 0.1 * feat_consistency_loss(f1, dx["feat2"]) \
            + 0.1 * prop_sym_matching_loss(dp["Recon"], dx["recon2"], dg["R"], dg["Tran"], dsym)
    elif piece == "torch":
        l = (dp["Tran"] - dg["Tran"]).abs().mean()
    if bwd:
        l.backward()
    return l


def zero():
    for v in list(dp.values()) + [f1]:
        if v.grad is not None:
            v.grad.zero_()


if "eager" in OPT:
    keep = step()
    eager = {k: v.grad.clone() for k, v in dp.items() if v.grad is not None}
    if "zero" in OPT:
        zero()
    if "drop" in OPT:                      # free the eager autograd graph: its AccumulateGrad nodes are bound to the default stream
        keep = float(keep)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
        if "zero" in OPT:
            zero()
torch.cuda.current_stream().wait_stream(side)
if "nosync" not in OPT:
    torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    l = step()
graph.replay()
torch.cuda.synchronize()
print("OK", piece, bwd, float(l))
