"""Development (GPU box): dW = dy^T x of the trainer's large layers, transposed-copy form against the transposed-LDS-read form."""
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
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    ts = sorted(ts[1:]); return ts[len(ts) // 2]


for rows, N, K in ((32896, 512, 512), (32896, 1024, 272), (32896, 256, 1024), (32896, 512, 272), (8224, 4608, 512), (2056, 4608, 512),
                   (32896, 256, 512), (8224, 2048, 128), (32896, 1024, 128)):
    if not ops.tn_split_ok(rows, N, K):
        print(rows, N, K, "not on the split path"); continue
    dy = torch.randn(rows, N, device=dev) * 1e-5
    x = torch.randn(rows, K, device=dev)
    sc = ops.absmax_scale(dy)
    res = {}
    for native in (True, False):
        ops.TN_NATIVE = native
        res[native] = timeit(lambda: ops.gemm_tn(dy, x, scale=sc)) * 1e3
    fl = 2.0 * rows * N * K
    print("rows %6d N %5d K %5d: native %7.1f us (%5.0f TF)   transposed copies %7.1f us (%5.0f TF)"
          % (rows, N, K, res[True], fl / res[True] / 1e6, res[False], fl / res[False] / 1e6), flush=True)
