"""Development (GPU box): in-kernel timestamps of the wide split GEMM: where a tile's time goes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tgpose_amd import _lib, ops
from _dev import use_dev_lib
lib = use_dev_lib()
dev = "cuda:0"
M, N, K, LD = 32896, 4096, 1292, 1292
mode = sys.argv[1] if len(sys.argv) > 1 else "split16"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ops.GEMM_MODE = mode
A = torch.randn(M, LD, device=dev); W = torch.randn(N, LD, device=dev) / K ** 0.5
C = torch.empty(M, N, device=dev)
WS = ops.split_w(W[:, :K].contiguous())
lib.tgp_debug_set_split_variant(variant)
for _ in range(3):
    ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, w_split=WS)
st = torch.zeros(4096, 5, dtype=torch.int64, device=dev)
lib.tgp_debug_set_split_stamps.argtypes = [ctypes.c_void_p]
lib.tgp_debug_set_split_stamps(ctypes.c_void_p(st.data_ptr()))
ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, w_split=WS)
torch.cuda.synchronize()
lib.tgp_debug_set_split_stamps(None)
s = st.cpu().numpy()
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = lambda x: x / 100.0
print("blocks", len(s), "kernel span %.1f us" % us(s[:, 3].max() - t0))
big = s[: 2048] if len(s) > 2048 else s
print("big tiles: prologue %.2f us  loop %.2f us  epilogue %.2f us  (medians)" % (
    np.median(us(big[:, 1] - big[:, 0])), np.median(us(big[:, 2] - big[:, 1])), np.median(us(big[:, 3] - big[:, 2]))))
print("percentiles epilogue us", np.percentile(us(big[:, 3] - big[:, 2]), [5, 25, 50, 75, 95]))
print("percentiles loop us", np.percentile(us(big[:, 2] - big[:, 1]), [5, 25, 50, 75, 95]))
if len(s) > 2048:
    sm = s[2048:]
    print("small tiles (%d): prologue %.2f us  loop %.2f us  epilogue (combine + store) %.2f us  start %.1f..%.1f" % (
        len(sm), np.median(us(sm[:, 1] - sm[:, 0])), np.median(us(sm[:, 2] - sm[:, 1])), np.median(us(sm[:, 3] - sm[:, 2])),
        us(sm[:, 0].min() - t0), us(sm[:, 0].max() - t0)))
starts = np.sort(us(s[:, 0] - t0))
print("block start times (us), every 256th:", starts[::256])
ends = np.sort(us(s[:, 3] - t0))
print("block end times (us), every 256th:", ends[::256])
