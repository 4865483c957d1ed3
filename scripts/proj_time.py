"""Times tgp_proj_planes (the HS layers' projection GEMM on its own kernel, csrc/hs_chain.hip) against the tile kernel on the same planes,
for the four projection shapes of Face_Enc at B = 32.    python scripts/proj_time.py [--rounds 5] [--reps 20] [--knobs 0,1]
(knobs, development library: 1 = no stores)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--knobs", default="")
    a = ap.parse_args()
    dev_lib = None
    if a.knobs:
        from _dev import use_dev_lib
        dev_lib = use_dev_lib()
    from tgpose_amd import ops
    dev = "cuda:0"
    for name, M, K, N in (("conv_1", 32896, 128, 1152), ("conv_2", 8224, 128, 2304), ("conv_3", 8224, 256, 2304), ("conv_4", 2048, 256, 4608), ("coarse 1", 8224, 512, 4608), ("coarse 2", 2048, 512, 4608)):
        gen = torch.Generator().manual_seed(1)
        A = torch.randn(M, K, generator=gen).to(dev)
        W, b = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev), (torch.randn(N, generator=gen).to(dev) if K < 512 else None)
        Ap, Wp, Ws, units = ops.planes_split(A, K=K), ops.planes_w(W), ops.split_w(W), ops.proj_pack(W)
        out = torch.empty(M, N, device=dev)
        tile = lambda: ops.linear_rows(A, W, bias=b, out=out, a_planes=Ap, w_planes=Wp, w_split=Ws)
        proj = lambda: ops.proj_planes(Ap, units, b, A, W, out=out)
        variants = [("tile kernel", tile, 0)] + [("tgp_proj_planes" + (" knobs=%d" % k if k else ""), proj, k)
                                                  for k in ([int(x) for x in a.knobs.split(",")] if a.knobs else [0])]
        for label, fn, knob in variants:
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
            mb = 4.0 * (M * K + N * K + M * N) / 1e6
            print("%-8s  M = %5d  K = %3d  N = %4d  %-24s median %6.1f us  min %6.1f   (%.0f MB: %.2f TB/s; %.0f TF-eq)" % (name, M, K, N, label, times[len(times) // 2], times[0], mb, mb / times[0], 2.0 * M * N * K / times[0] / 1e6), flush=True)


if __name__ == "__main__":
    main()
