"""Times tgp_dec_l1 (the decoder's first conv on the factored form, csrc/dec_fused.hip) alone at the benchmark's shape on random operands.

    python scripts/dec_l1_time.py [--rounds 5] [--reps 20] [--knobs 0,1,2,3]     # knobs (development library): 1 = no stores, 2 = no gathers
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=1028)
    ap.add_argument("--knobs", default="")
    a = ap.parse_args()
    dev_lib = None
    if a.knobs:
        from _dev import use_dev_lib
        dev_lib = use_dev_lib()
    from tgpose_amd import ops
    dev = "cuda:0"
    B, N, K = a.batch, a.points, 268
    M = B * N
    gen = torch.Generator().manual_seed(2)
    d = lambda t: t.contiguous().to(dev)
    fine = torch.randn(M, 272, generator=gen)
    fine[:, K:] = 0
    W = torch.randn(512, 272, generator=gen) / K ** 0.5
    W[:, K:] = 0
    n1, n2 = B * (N // 4), B * (N // 16)
    P1, P2 = d(torch.randn(n1, 512, generator=gen)), d(torch.randn(n2, 512, generator=gen))
    idx1 = d(torch.sort(torch.randint(0, n1, (M,), generator=gen, dtype=torch.int32))[0])
    idx2 = d(torch.sort(torch.randint(0, n2, (M,), generator=gen, dtype=torch.int32))[0])
    bias, scale, shift = d(torch.randn(512, generator=gen) * 0.1), d(torch.rand(512, generator=gen) + 0.5), d(torch.randn(512, generator=gen) * 0.1)
    rb = d(torch.randn(B, 512, generator=gen) * 0.1)
    pl = ops.planes_split(d(fine), K=K, kt=17)
    wp = ops.heads_planes_w(d(W))
    h1 = ops.Planes(M, 512, dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lambda: ops.dec_l1(pl, wp, P1, idx1, P2, idx2, bias, scale, shift, rb, N, h1, flag)
    for k in ([int(x) for x in a.knobs.split(",")] if a.knobs else [0]):
        if dev_lib is not None:
            dev_lib.tgp_debug_set_dec_l1_knobs(k)
        times = []
        for _ in range(a.rounds):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / a.reps * 1e3)
        times.sort()
        print("dec_l1  knobs=%d  median %7.1f us   min %7.1f" % (k, times[len(times) // 2], times[0]), flush=True)


if __name__ == "__main__":
    main()
