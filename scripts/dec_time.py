"""The fused decoder kernel (tgp_dec_fused) against the launches it replaces (three tile GEMMs on planes + tgp_rows_out) on random operands at
the benchmark's shape: agreement and time.   python scripts/dec_time.py [--batch 32] [--points 1028]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgpose_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=1028)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda:0"
    B, N = a.batch, a.points
    M = B * N
    gen = torch.Generator().manual_seed(3)
    d = lambda t: t.contiguous().to(dev)
    H1 = d(torch.relu(torch.randn(M, 512, generator=gen)))
    Ws = [d(torch.randn(n, k, generator=gen) / k ** 0.5) for n, k in ((512, 512), (256, 512), (128, 256))]
    vecs = [(d(torch.randn(n, generator=gen) * 0.1), d(torch.rand(n, generator=gen) + 0.5), d(torch.randn(n, generator=gen) * 0.1)) for n in (512, 256, 128)]
    w5, b5 = d(torch.randn(3, 128, generator=gen) / 11.0), d(torch.randn(3, generator=gen))
    order = d(torch.stack([torch.randperm(N, generator=gen) for _ in range(B)]))
    h1p = ops.planes_split(H1, K=512)
    units = ops.dec_pack(*Ws)
    flag = torch.zeros(1, device=dev, dtype=torch.int32)
    wsp = [ops.split_w(w) for w in Ws]
    wpl = [ops.planes_w(w) for w in Ws]

    def chain():
        xp = h1p
        x = torch.empty(M, 128, device=dev)
        planes = [ops.Planes(M, 512, dev), ops.Planes(M, 256, dev), None]
        for i in range(3):
            n, k = Ws[i].shape
            ops.gemm(None, Ws[i], x if i == 2 else None, M=M, N=n, K=k, lda=0, ldw=k, ldc=n, bias=vecs[i][0], scale=vecs[i][1], shift=vecs[i][2],
                     act=1, w_split=wsp[i], a_planes=xp, w_planes=wpl[i], c_planes=planes[i], range_flag=flag)
            xp = planes[i]
        return ops.rows_out(x.view(B, N, 128), w5, b5, order)

    def fused():
        return ops.dec_fused(h1p, units, vecs, w5, b5, order, N, flag).view(B, N, 3)

    tile = ops._routes_to_big_tile(M, 128, 1, True)
    got = fused()
    want = chain() if tile else got
    torch.cuda.synchronize()
    # fp64 restatement
    x = H1.double()
    for i in range(3):
        x = torch.relu((x @ Ws[i].double().t() + vecs[i][0].double()) * vecs[i][1].double() + vecs[i][2].double())
    y = x @ w5.double().t() + b5.double()
    ref = torch.empty(B, N, 3, dtype=torch.float64, device=dev)
    ref.scatter_(1, order.unsqueeze(-1).expand(-1, -1, 3), y.view(B, N, 3))
    sc = ref.abs().max().item()
    print("flag %d; |fused - chain| max %.3g; |chain - fp64| %.3g; |fused - fp64| %.3g  (output scale %.3g)"
          % (int(flag.item()), (got - want).abs().max().item(), (want.double() - ref).abs().max().item(), (got.double() - ref).abs().max().item(), sc))
    for name, fn in ((("chain (3 tile GEMMs + rows_out)", chain),) if tile else ()) + (("fused", fused),):
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / a.reps * 1e3)
        print("%-34s %7.1f us (min of 3: %s)" % (name, min(ts), ", ".join("%.1f" % t for t in ts)))


if __name__ == "__main__":
    main()
