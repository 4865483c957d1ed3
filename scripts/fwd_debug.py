"""Development: per-key errors of the eval forward with the training keys (concat path) against the oracle, normal and conv_4-scaled weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import PoseNet9D, seeded_state_dict, FLAGS
from oracle import posenet_ref as PR
from tests.util import synth_points
DEV = "cuda:0"
for scale4 in (1.0, 1e6):
    sd = seeded_state_dict(14)
    for k in ("weights", "bias", "STE_layer.weight"):
        sd["face_all.encoder.conv_4." + k] = sd["face_all.encoder.conv_4." + k] * scale4
    B, N = 2, 512
    pts, obj = synth_points(B, N, 14)
    torch.manual_seed(2)
    i1 = torch.randperm(N)[: N // 4]
    sample = (i1, torch.randperm(i1.numel())[: i1.numel() // 4])
    with torch.no_grad():
        want, inter = PR.posenet_forward(sd, pts, obj, sample_idx=sample, train_keys=True, mode="exact", want_intermediates=True)
    net = PoseNet9D()
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    for train in (1, 0):
        FLAGS.train = train
        with torch.no_grad():
            probe = {}
            got = net(pts.to(DEV), obj.to(DEV), sample_idx=sample, inject={k: v.int() for k, v in inter["indices"].items()})
        FLAGS.train = 0
        print("scale4=%g train_keys=%d" % (scale4, train))
        for k, v in want.items():
            if k not in got:
                continue
            a = got[k].cpu()
            print("   %-12s err %.3e  scale %.3e  finite %s" % (k, (a - v).abs().max().item(), v.abs().max().item(), bool(torch.isfinite(a).all())))
        if train:
            f = got["feat"].cpu()
            w = want["feat"]
            for lo, hi, nm in ((0, 128, "fm0"), (128, 256, "fm1"), (256, 512, "fm2"), (512, 768, "fm3"), (768, 1280, "fm4"), (1280, 1286, "tail")):
                print("      feat[%s] err %.3e scale %.3e" % (nm, (f[:, :, lo:hi] - w[:, :, lo:hi]).abs().max().item(), w[:, :, lo:hi].abs().max().item()))
