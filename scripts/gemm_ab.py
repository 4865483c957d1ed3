"""Development A/B (GPU box): production tgp_gemm_f32 vs the micro-benchmark variants on identical operands."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import _lib, ops

from _dev import use_dev_lib
lib = use_dev_lib()
fn = lib.tgp_debug_gemm_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
LD = 1292

def timeit(f, reps=3, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts = sorted(ts[1:]); return ts[len(ts)//2]

for (M, N, K) in ((32768, 4096, 1280), (32896, 4096, 1292), (32896, 512, 1292), (32896, 1152, 128)):
    A = torch.randn(M, LD, device=dev); W = torch.randn(N, LD, device=dev) / K ** 0.5
    C = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev); scale = torch.rand(N, device=dev) + .5; shift = torch.randn(N, device=dev)
    flops = 2.0 * M * N * K
    res = {}
    if M % 256 == 0 and K % 16 == 0:
        for v in (13, 14):
            res["variant %d" % v] = timeit(lambda: fn(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, LD, v, None, stream))
    res["prod plain"] = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N))
    ref = (A[:300, :K].double() @ W[:, :K].double().t())
    e32 = (C[:300].double() - ref).abs().max().item()
    WS = ops.split_bf16(W[:, :K].contiguous())
    C.zero_()
    es = 0.0
    for sv, nm in ((0, "single buffer"), (1, "double buffer")):
        lib.tgp_debug_set_split_variant(sv)
        C.zero_()
        res["split " + nm] = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, w_split=WS))
        es = max(es, (C[:300].double() - ref).abs().max().item())
    lib.tgp_debug_set_split_variant(1)
    res["split bias+bn+relu"] = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, bias=bias, scale=scale, shift=shift, act=1, w_split=WS))
    print("   max |err| vs fp64: fp32-MFMA %.3e   bf16x3-split %.3e   (|ref| max %.2f)" % (e32, es, ref.abs().max().item()))
    res["prod bias+bn+relu"] = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, bias=bias, scale=scale, shift=shift, act=1))
    keys = torch.zeros(M // 1028 + 1, N, dtype=torch.int32, device=dev)
    res["prod colmax only"] = timeit(lambda: ops.gemm(A, W, None, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=0, bias=bias, scale=scale, shift=shift, act=1, colmax_keys=keys, rows_per_obj=1028))
    print("M=%d N=%d K=%d" % (M, N, K))
    for k, t in res.items():
        print("   %-20s %.3f ms  %.1f TF" % (k, t, flops / t / 1e9))
