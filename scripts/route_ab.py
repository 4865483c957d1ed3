"""Development A/B (GPU box): 512-thread (two workgroups per CU, 256 x 128 tiles) against 1024-thread (256 x 256 tiles) routing of
the fp16-split GEMM on the forward's M = 32896 shapes; variant bit 16 forces the 512-thread form, bit 32 forbids it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _dev import use_dev_lib
lib = use_dev_lib()
from tgpose_amd import ops
dev = "cuda:0"


def timeit(f, reps=5, rounds=7):
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
for (M, N, K) in ((32896, 512, 512), (32896, 512, 268), (32896, 1152, 128), (32896, 256, 512), (8224, 2304, 128), (8224, 2304, 256),
                  (2048, 4608, 256), (8224, 4608, 512), (2048, 4608, 512)):
    LD = (K + 15) // 16 * 16
    A = torch.randn(M, LD, device=dev); W = torch.randn(N, LD, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev)
    WS = ops.split_f16(W[:, :K].contiguous())
    line = "M=%d N=%d K=%d:" % (M, N, K)
    for name, v in (("default", 7), ("force512", 2 | 16), ("forbid512", 2 | 32)):
        lib.tgp_debug_set_split_variant(v)
        t = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, w_split=WS))
        line += "  %s %.1f us" % (name, t * 1e3)
    lib.tgp_debug_set_split_variant(7)
    print(line, flush=True)
