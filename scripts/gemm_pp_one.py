"""GPU box: one launch shape of the pre-split GEMM, repeated (for rocprofv3 --pmc passes): python scripts/gemm_pp_one.py [M N K cfg reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import ops

M, N, K, cfg, reps = [int(x) for x in (sys.argv[1:6] + ["32896", "512", "512", "5", "20"][len(sys.argv) - 1:])]
dev = "cuda:0"
torch.manual_seed(0)
A = torch.randn(M, K, device=dev)
W = (torch.randn(N, K, device=dev) / K ** 0.5).contiguous()
Ws, Wp, Ap = ops.split_w(W), ops.planes_w(W), ops.planes_split(A, K=K)
C = torch.empty(M, N, device=dev)
bias = torch.randn(N, device=dev)
for _ in range(reps):
    ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, w_split=Ws, a_planes=Ap, w_planes=Wp, pp_config=cfg, bias=bias,
             scale=torch.ones(N, device=dev), shift=torch.zeros(N, device=dev), act=1)
torch.cuda.synchronize()
print("done")
