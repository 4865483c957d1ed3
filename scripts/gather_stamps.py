"""Development (GPU box): in-kernel timestamps of the factored wide layer's fine GEMM (K = 268) with 0 / 1 / 2 gathered residuals."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tgpose_amd import _lib, ops
from _dev import use_dev_lib
lib = use_dev_lib()
dev = "cuda:0"
B, Np, N1, N2 = 32, 1028, 257, 64
M, N, K, LD = B * Np, 4096, 268, 272
A = torch.randn(M, LD, device=dev); W = torch.randn(N, LD, device=dev) / K ** 0.5
W[:, K:] = 0
C = torch.empty(M, 3072, device=dev)
WS = ops.split_w(W)
P1 = torch.randn(B * N1, 4608, device=dev); P2 = torch.randn(B * N2, 4608, device=dev)
g = torch.Generator().manual_seed(0)
n1 = torch.sort(torch.randint(0, N1, (B, Np), generator=g), dim=1)[0]
n2 = (n1 * N2 // N1)
rows = torch.arange(B).unsqueeze(1)
i1 = (n1 + rows * N1).int().to(dev).contiguous(); i2 = (n2 + rows * N2).int().to(dev).contiguous()
bias = torch.randn(N, device=dev); sc = torch.rand(N, device=dev) + 0.5; sh = torch.randn(N, device=dev); sl = torch.zeros(N, device=dev)
lib.tgp_debug_set_split_stamps.argtypes = [ctypes.c_void_p]
for mode in (0, 1, 2):
    keys = torch.zeros(B, 1024, device=dev, dtype=torch.int32)
    kw = dict(M=M, N=N, K=K, lda=LD, ldw=LD, ldc=3072, bias=bias, scale=sc, shift=sh, act=1, slope_vec=sl, colmax_keys=keys, cm_cols=1024,
              c_col0=1024, rows_per_obj=Np, w_split=WS)
    if mode >= 1: kw["gather1"] = (P1, 4608, i1)
    if mode >= 2: kw["gather2"] = (P2, 4608, i2)
    for _ in range(3):
        ops.gemm(A, W, C, **kw)
    st = torch.zeros(4096, 5, dtype=torch.int64, device=dev)
    lib.tgp_debug_set_split_stamps(ctypes.c_void_p(st.data_ptr()))
    ops.gemm(A, W, C, **kw)
    torch.cuda.synchronize()
    lib.tgp_debug_set_split_stamps(None)
    s = st.cpu().numpy(); s = s[s[:, 0] > 0]
    us = lambda x: x / 100.0
    big = s[:2048]
    print("gathers=%d: span %.1f us; big tiles: prologue %.2f loop %.2f epilogue %.2f us (medians)" % (
        mode, us(s[:, 3].max() - s[:, 0].min()), np.median(us(big[:, 1] - big[:, 0])), np.median(us(big[:, 2] - big[:, 1])),
        np.median(us(big[:, 3] - big[:, 2]))))
