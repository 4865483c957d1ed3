"""Development (GPU box): time and check the M <= 32 weight-streaming GEMM on the forward's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import ops
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
flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)
for (M, N, K) in ((32, 1024, 2048), (32, 5000, 1024), (32, 1286, 5000), (32, 512, 1292), (32, 256, 256), (7, 1286, 5000), (32, 8, 256)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    C = ops.linear_rows(A, W, bias=b)
    ref = A.double() @ W.double().t() + b.double()
    err = (C.double() - ref).abs().max().item()
    t_hot = timeit(lambda: ops.linear_rows(A, W, bias=b, out=C))
    def cold():
        flush.zero_(); ops.linear_rows(A, W, bias=b, out=C)
    t_cold = timeit(cold, reps=1) - timeit(lambda: flush.zero_(), reps=1)
    print("M=%d N=%d K=%d  err %.2e  hot %.1f us (%.0f GB/s)  cold ~%.1f us" % (M, N, K, err, t_hot * 1e3, N * K * 4 / t_hot / 1e6, t_cold * 1e3), flush=True)
