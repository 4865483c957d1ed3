"""Development A/B (GPU box): the operand-split GEMM kernels (bf16x3 / fp16x2) at each register-prefetch depth,
on the forward's GEMM shapes, in one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import _lib, ops

from _dev import use_dev_lib
lib = use_dev_lib()
dev = "cuda:0"

def timeit(f, reps=3, rounds=7):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts = sorted(ts[1:]); return ts[len(ts)//2]

shapes = ((32896, 4096, 1292, 1292), (32896, 512, 1292, 1292), (32896, 1152, 128, 128), (32896, 512, 512, 512),
          (8224, 2304, 128, 128), (8224, 2304, 256, 256), (32896, 256, 1024, 1024))
for (M, N, K, LD) in shapes:
    A = torch.randn(M, LD, device=dev); W = torch.randn(N, LD, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev)
    flops = 2.0 * M * N * K
    line = "M=%d N=%d K=%d:" % (M, N, K)
    for mode, split in (("split", ops.split_bf16), ("split16", ops.split_f16)):
        ops.GEMM_MODE = mode
        WS = split(W[:, :K].contiguous())
        for v in ((1,) if mode == "split" else (2, 66)):
            lib.tgp_debug_set_split_variant(v)
            t = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, w_split=WS))
            line += "  %s/v%d %.3f ms %.0f TF" % (mode, v, t, flops / t / 1e9)
            if mode == "split16":
                ref = A[-300:, :K].double() @ W[:, :K].double().t()
                line += " tail-rows err %.1e" % ((C[-300:].double() - ref).abs().max().item() / ref.abs().max().item())
    lib.tgp_debug_set_split_variant(7)
    print(line, flush=True)
