"""Development (GPU box): LDS-staged graph convolution vs the gather-from-L2 kernel: bit identity and time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import _lib, ops
from _dev import use_dev_lib
lib = use_dev_lib()
dev = "cuda:0"
def timeit(f, reps=5, rounds=7):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts = sorted(ts[1:]); return ts[len(ts)//2]
gen = torch.Generator().manual_seed(0)
for (B, n, k, C) in ((32, 1028, 20, 128), (32, 257, 20, 256), (32, 64, 8, 512), (3, 300, 12, 256), (256, 1028, 20, 128)):
    xyz = (torch.randn(B, n, 3, generator=gen) * 0.1).to(dev)
    idx = torch.stack([torch.stack([torch.randperm(n, generator=gen)[:k] for _ in range(n)]) for _ in range(min(B, 4))]).int()
    idx = idx.repeat((B + 3) // 4, 1, 1)[:B].contiguous().to(dev)
    proj = torch.randn(B, n, 9 * C, generator=gen).to(dev)
    sdn = ops.normalize_dirs(torch.randn(3, 7 * C, generator=gen).to(dev))
    res = {}
    for mode in (0, 1):
        lib.tgp_debug_set_gconv_lds(mode)
        out = ops.gconv_hs(xyz, idx, proj, sdn, 7, C)
        res[mode] = (out.clone(), timeit(lambda: ops.gconv_hs(xyz, idx, proj, sdn, 7, C)))
    print("B=%d n=%d k=%d C=%d  L2-gather %.1f us  LDS-staged %.1f us  identical %s" % (
        B, n, k, C, res[0][1] * 1e3, res[1][1] * 1e3, torch.equal(res[0][0], res[1][0])), flush=True)
lib.tgp_debug_set_gconv_lds(1)

print("ORL pooling (orl_rowbias):")
for (B, n, k, C) in ((32, 1028, 20, 128), (32, 257, 20, 256), (32, 64, 8, 512), (5, 300, 12, 256)):
    idx = torch.stack([torch.stack([torch.randperm(n, generator=gen)[:k] for _ in range(n)]) for _ in range(min(B, 4))]).int()
    idx = idx.repeat((B + 3) // 4, 1, 1)[:B].contiguous().to(dev)
    feat = torch.randn(B, n, C, generator=gen).to(dev)
    w2t = torch.randn(C, C, generator=gen).to(dev)
    res = {}
    for mode in (0, 1):
        lib.tgp_debug_set_orl_lds(mode)
        out = ops.orl_rowbias(feat, idx, w2t)
        res[mode] = (out.clone(), timeit(lambda: ops.orl_rowbias(feat, idx, w2t)))
    print("B=%d n=%d k=%d C=%d  L2-gather %.1f us  LDS-staged %.1f us  identical %s" % (
        B, n, k, C, res[0][1] * 1e3, res[1][1] * 1e3, torch.equal(res[0][0], res[1][0])), flush=True)
lib.tgp_debug_set_orl_lds(1)
