"""Times tgp_hs_chain (an HS layer's last GEMM + the next layer's projection in one launch) against the two tile-kernel launches it
replaces, at the benchmark's two shapes (B = 32): conv_0 -> conv_1 (M = 32896, 132 -> 128 -> 1152) and conv_2 -> conv_3 (M = 8224,
256 -> 256 -> 2304).

    python scripts/hs_chain_time.py [--rounds 5] [--reps 20]
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
    ap.add_argument("--knobs", default="", help="development library: timing-only variants, e.g. 0,1,2,3,4 (1 = no stores of the projection, "
                                                "2 = none of the intermediate, 4 = no projection at all; results are garbage)")
    a = ap.parse_args()
    dev_lib = None
    if a.knobs:
        from _dev import use_dev_lib
        dev_lib = use_dev_lib()
    from tgpose_amd import ops
    dev = "cuda:0"
    for B, n, K1, N1, N2, bn in ((32, 1028, 132, 128, 1152, False), (32, 257, 256, 256, 2304, True)):
        gen = torch.Generator().manual_seed(1)
        M = B * n
        g = lambda t: t.to(dev)
        A = g(torch.randn(M, K1, generator=gen))
        W1, W2 = g(torch.randn(N1, K1, generator=gen) / K1 ** 0.5), g(torch.randn(N2, N1, generator=gen) / N1 ** 0.5)
        rowbias, b2, res1 = g(torch.randn(B, N1, generator=gen)), g(torch.randn(N2, generator=gen)), g(torch.randn(M, N1, generator=gen))
        wide = g(torch.randn(M, 9 * N1, generator=gen))
        res2 = wide[:, 8 * N1:] if bn else None
        scale, shift = (g(torch.rand(N1, generator=gen) + 0.5), g(torch.randn(N1, generator=gen))) if bn else (None, None)
        Ap, W1p, W2p, W1s, W2s = ops.planes_split(A, K=K1), ops.planes_w(W1), ops.planes_w(W2), ops.split_w(W1), ops.split_w(W2)
        units = ops.hs_chain_pack(W1, W2)
        c1 = torch.empty(M, N1, device=dev)
        c2 = torch.empty(M, N2, device=dev)
        pl = ops.Planes(M, N1, dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)

        def two():
            ops.linear_rows(A, W1, out=c1, rowbias=rowbias, rows_per_obj=n, res1=res1, res2=res2, scale=scale, shift=shift, act=1, slope=0.0,
                            a_planes=Ap, w_planes=W1p, c_planes=pl, w_split=W1s)
            ops.linear_rows(c1, W2, bias=b2, out=c2, a_planes=pl, w_planes=W2p, w_split=W2s)

        def one():
            ops.hs_chain(Ap, units, c1, b2, flag, rowbias=rowbias, rows_per_obj=n, res1=res1, res2=res2, scale1=scale, shift1=shift, relu=True,
                         c1_planes=pl, c2=c2)

        variants = [("two tile-kernel launches", two, 0)] + [("tgp_hs_chain" + (" knobs=%d" % k if k else ""), one, k)
                                                               for k in ([int(x) for x in a.knobs.split(",")] if a.knobs else [0])]
        for name, fn, knob in variants:
            if dev_lib is not None:
                dev_lib.tgp_debug_set_hs_chain_knobs(knob)
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
            print("M = %5d  %3d -> %3d -> %4d   %-26s median %7.1f us   min %7.1f" % (M, K1, N1, N2, name, times[len(times) // 2], times[0]), flush=True)
        assert int(flag.item()) == 0


if __name__ == "__main__":
    main()
