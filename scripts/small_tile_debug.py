"""Development: the 64 x 128 split tile (gemm_split256_kernel) against fp64 on the shapes it serves, with operands inside and outside
fp16's range (the range guard's exact-fp32 recomputation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import ops
dev = "cuda:0"
for (M, N, K) in [(1024, 512, 512), (1024, 256, 512), (32896, 128, 128), (8224, 256, 256), (2048, 512, 512), (32896, 128, 256), (100, 130, 36)]:
    for a_mag, huge_rows in [(1.0, 0), (3e5, 0), (1.0, 5)]:
        gen = torch.Generator().manual_seed(M + K)
        A = torch.randn(M, K, generator=gen) * a_mag
        if huge_rows:
            A[torch.randint(0, M, (huge_rows,), generator=gen)] *= 1e6
        W = torch.randn(N, K, generator=gen) / K ** 0.5
        b = torch.randn(N, generator=gen)
        ref = A.double() @ W.double().t() + b.double()
        dA, dW = A.to(dev), W.to(dev)
        out = ops.linear_rows(dA, dW, bias=b.to(dev), w_split=ops.split_w(dW))
        torch.cuda.synchronize()
        err = (out.cpu().double() - ref).abs()
        rel = (err / ref.abs().max(dim=1, keepdim=True)[0]).max().item()
        bad_rows = (err.max(dim=1)[0] > 1e-4 * ref.abs().max(dim=1)[0]).nonzero().flatten()
        print("M=%d N=%d K=%d a_mag=%g huge_rows=%d: max err / row scale %.3e, finite %s, bad rows %d %s"
              % (M, N, K, a_mag, huge_rows, rel, bool(torch.isfinite(out).all()), bad_rows.numel(), bad_rows[:8].tolist()), flush=True)

print("---- full epilogue")
for (M, N, K, rpo) in [(1024, 512, 512, 512), (1024, 128, 128, 512), (1024, 256, 512, 512), (4112, 128, 128, 1028), (2056, 128, 128, 257), (8224, 256, 256, 257), (2048, 512, 512, 64)]:
    for a_mag in (1.0, 3e5):
        gen = torch.Generator().manual_seed(M + K + 1)
        A = torch.randn(M, K, generator=gen) * a_mag
        W = torch.randn(N, K, generator=gen) / K ** 0.5
        nobj = (M + rpo - 1) // rpo
        bias, scale, shift = torch.randn(N, generator=gen), torch.rand(N, generator=gen) + 0.5, torch.randn(N, generator=gen)
        rowbias = torch.randn(nobj, N, generator=gen)
        res1, res2 = torch.randn(M, N, generator=gen), torch.randn(M, 9 * N, generator=gen)
        v = A.double() @ W.double().t() + bias.double() + rowbias.double().repeat_interleave(rpo, dim=0)[:M] + res1.double() + res2[:, 8 * N:].double()
        v = v * scale.double() + shift.double()
        ref = torch.where(v > 0, v, v * 0.0)
        dA, dW = A.to(dev), W.to(dev)
        out = torch.empty(M, N, device=dev)
        ops.linear_rows(dA, dW, out=out, bias=bias.to(dev), rowbias=rowbias.to(dev), rows_per_obj=rpo, res1=res1.to(dev),
                        res2=res2.to(dev)[:, 8 * N:], scale=scale.to(dev), shift=shift.to(dev), act=1, slope=0.0, w_split=ops.split_w(dW))
        torch.cuda.synchronize()
        err = (out.cpu().double() - ref).abs()
        print("M=%d N=%d K=%d rpo=%d a_mag=%g: max err %.3e of scale %.3e, finite %s, bad rows %s"
              % (M, N, K, rpo, a_mag, err.max().item(), ref.abs().max().item(), bool(torch.isfinite(out).all()),
                 (err.max(dim=1)[0] > 1e-4 * ref.abs().max()).nonzero().flatten()[:8].tolist()), flush=True)
