"""Development (GPU box): run only the wide split GEMM a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import ops
dev = "cuda:0"
M, N, K, LD = 32896, 4096, 1292, 1292
A = torch.randn(M, LD, device=dev); W = torch.randn(N, LD, device=dev) / K ** 0.5
C = torch.empty(M, N, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "split"
ops.GEMM_MODE = mode
WS = ops.split_w(W[:, :K].contiguous())
for _ in range(6):
    ops.gemm(A, W, C, M=M, N=N, K=K, lda=LD, ldw=LD, ldc=N, w_split=WS)
torch.cuda.synchronize()
print("done", mode)
