"""GPU box, development build: where a workgroup of the pre-split GEMM kernel spends its time (wall-clock stamps at entry, first
stage landed, K loop done, epilogue done) on the small launches of the eval forward (B = 32, N = 1028).
    python scripts/gemm_pp_stamps.py [out.txt]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _dev import use_dev_lib
lib = use_dev_lib()
import torch
from tgpose_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
B, NP = 32, 1028
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout


def say(*a):
    print(*a, file=out, flush=True)
    if out is not sys.stdout:
        print(*a, flush=True)


SHAPES = [
    ("conv_1 last", B * NP, 128, 128, "layer2", 5),
    ("conv_4 last", B * 64, 512, 512, "layer2_64", 5),
    ("dec 256-128", B * NP, 128, 256, "bn", 5),
    ("dec 512-512", B * NP, 512, 512, "bn", 5),
    ("conv_1 proj", B * NP, 1152, 128, "bias", 4),
    ("coarse 1", B * 257, 4608, 512, "plain", 3),
]
lib.tgp_debug_set_pp_stamps.argtypes = [ctypes.c_void_p]
lib.tgp_debug_set_pp_stamps.restype = ctypes.c_int
stamps = torch.zeros((1 << 18) + (1 << 17), device=dev, dtype=torch.int64)      # 64 K workgroups x 4 wall stamps, then x 2 shader-clock stamps
assert lib.tgp_debug_set_pp_stamps(stamps.data_ptr()) == 0
lib.tgp_debug_set_pp_knobs.argtypes = [ctypes.c_int]
KNOBS = [int(x) for x in os.environ.get("TGP_PP_KNOBS", "0").split(",")]     # timing-only variants (garbage results): see csrc/gemm_pp.hip
for name, M, N, K, fl, cfg in SHAPES:
    A = torch.randn(M, K, device=dev)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).contiguous()
    Ws, Wp, Ap = ops.split_w(W), ops.planes_w(W), ops.planes_split(A, K=K)
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldc=N, w_split=Ws)
    n_obj = {"layer2_64": 64}.get(fl, NP if M == B * NP else 257)
    if fl != "plain":
        kw["bias"] = torch.randn(N, device=dev)
    if fl in ("bn", "layer2", "layer2_64"):
        kw.update(scale=torch.rand(N, device=dev) + 0.5, shift=torch.randn(N, device=dev), act=1)
    if fl in ("layer2", "layer2_64"):
        kw.update(rowbias=torch.randn(M // n_obj, N, device=dev), rows_per_obj=n_obj, res1=torch.randn(M, N, device=dev), ldr1=N,
                  res2=torch.randn(M, N, device=dev), ldr2=N)
    C = torch.empty(M, N, device=dev)
    Cp = ops.Planes(M, N, dev)
    for with_planes, knob in [(False, k) for k in KNOBS] + ([(True, 0)] if KNOBS == [0] else []):
        assert lib.tgp_debug_set_pp_knobs(knob) == 0
        for _ in range(3):
            stamps.zero_()
            torch.cuda.synchronize()
            ops.gemm(A, W, C, a_planes=Ap, w_planes=Wp, pp_config=cfg, c_planes=Cp if with_planes else None, **kw)
            torch.cuda.synchronize()
        raw = stamps.cpu()
        s = raw[: 1 << 18].view(-1, 4)
        cyc = raw[1 << 18:].view(-1, 2)
        live = s[:, 3] != 0
        mhz = float(((cyc[live, 1] - cyc[live, 0]).double() / ((s[live, 3] - s[live, 0]).double() / 100.0)).median())
        s = s[live].double() / 100.0           # us
        t0 = s[:, 0].min()
        med = lambda v: float(v.median())
        say("%-12s M=%6d N=%5d K=%4d cfg %d knobs %d planes-out %d | %4d workgroups, span %6.1f us; last entry at %5.1f; per workgroup (median / max): "
            "to first stage %5.2f / %5.2f, K loop %5.2f / %5.2f, epilogue %5.2f / %5.2f, total %5.2f / %5.2f; shader clock %4.0f MHz"
            % (name, M, N, K, cfg, knob, with_planes, s.shape[0], float(s[:, 3].max() - t0), float(s[:, 0].max() - t0),
               med(s[:, 1] - s[:, 0]), float((s[:, 1] - s[:, 0]).max()), med(s[:, 2] - s[:, 1]), float((s[:, 2] - s[:, 1]).max()),
               med(s[:, 3] - s[:, 2]), float((s[:, 3] - s[:, 2]).max()), med(s[:, 3] - s[:, 0]), float((s[:, 3] - s[:, 0]).max()), mhz))
