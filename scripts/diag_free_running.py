"""Diagnostic (GPU box): how far does a free-running HIP forward drift from the CPU oracle, given that
feature-space kNN is ill-conditioned (distance quantisation from cancellation -> near-ties)?"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import synth_points
from oracle import posenet_ref as PR
from tgpose_amd import PoseNet9D, seeded_state_dict, FLAGS

dev = "cuda:0"
for B, N, seed in ((4, 1028, 11), (2, 1024, 12), (3, 512, 13)):
    sd = seeded_state_dict(seed)
    net = PoseNet9D(); net.load_state_dict(sd); net = net.to(dev).eval()
    pts, obj = synth_points(B, N, seed)
    torch.manual_seed(seed)
    i1 = torch.randperm(N)[: N // 4]
    sample = (i1, torch.randperm(i1.numel())[: i1.numel() // 4])
    with torch.no_grad():
        want, inter = PR.posenet_forward(sd, pts, obj, sample_idx=sample, train_keys=True, mode="exact", want_intermediates=True)
        want_t, inter_t = PR.posenet_forward(sd, pts, obj, sample_idx=sample, train_keys=True, mode="torch", want_intermediates=True)
    FLAGS.train = 1
    rec = {}
    free = net(pts.to(dev), obj.to(dev), sample_idx=sample, record=rec)
    print("case", B, N, seed)
    for name, idx in inter["indices"].items():
        got = rec[name].cpu().long()
        got = (got.unsqueeze(-1) if got.dim() == 2 else got)[..., : idx.shape[-1]]
        ordered = (got == idx).all(-1).float().mean().item()
        sets = (got.sort(-1)[0] == idx.sort(-1)[0]).all(-1).float().mean().item()
        it = inter_t["indices"][name]
        sets_t = (it.sort(-1)[0] == idx.sort(-1)[0]).all(-1).float().mean().item()
        print("  %-36s ordered-equal %.4f  set-equal %.4f | oracle exact-vs-torch-mode set-equal %.4f" % (name, ordered, sets, sets_t))
    for k in want:
        d = (free[k].cpu() - want[k]).abs()
        dt = (want_t[k] - want[k]).abs()
        print("  out %-12s max|hip-oracle| %.3e (mean %.2e) | max|oracle torch-mode - exact| %.3e  scale %.2f" % (k, d.max().item(), d.mean().item(), dt.max().item(), want[k].abs().max().item()))
