"""Development (GPU box): asymptotic (long-reduction) rate of the NT split tile kernel against the TN transposed-read kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import ops, _lib
dev = "cuda:0"


def timeit(f, reps=3, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts = sorted(ts[1:]); return ts[len(ts) // 2]


ops.GEMM_MODE = "split16"
for (M, N, K) in ((8224, 4608, 512), (8224, 4608, 2048), (8224, 4608, 8192), (32896, 512, 512), (32896, 512, 4096)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev)
    WS = ops.split_f16(W)
    t = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, w_split=WS)) * 1e3
    print("NT  M %6d N %5d K %5d: %8.1f us  %6.0f TF" % (M, N, K, t, 2.0 * M * N * K / t / 1e6), flush=True)
for (rows, N, K) in ((8192, 4608, 8224), (2048, 4608, 8224), (32896, 512, 512)):
    a = torch.randn(rows, N, device=dev); b = torch.randn(rows, K, device=dev)
    sc = ops.absmax_scale(a)
    Z, chunk = 1, (rows + 31) // 32 * 32
    parts = torch.empty(Z, N, K, device=dev)
    f = lambda: ops.check(_lib.lib().tgp_gemm_tn_split(ops._p(a), N, ops._p(b), K, rows, N, K, ops._p(sc), Z, chunk, ops._p(parts), ops._stream(a)), "tn")
    t = timeit(f) * 1e3
    print("TN  rows %6d N %5d K %5d (one chunk): %8.1f us  %6.0f TF" % (rows, N, K, t, 2.0 * rows * N * K / t / 1e6), flush=True)
