"""GPU box: the pre-split GEMM kernel (csrc/gemm_pp.hip, every tile shape) against the in-loop-split kernels of csrc/gemm.hip on the
eval forward's GEMM launches (B = 32, N = 1028): results must be bit-identical; times are medians of interleaved rounds in one process.
    python scripts/gemm_pp_ab.py [out.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


# (name, M, N, K, flavour): flavours pick the epilogue of the forward's launch
SHAPES = [
    ("conv_1 proj", B * NP, 1152, 128, "bias"),
    ("conv_2 proj", B * 257, 2304, 128, "bias"),
    ("conv_3 proj", B * 257, 2304, 256, "bias"),
    ("conv_4 proj", B * 64, 4608, 256, "bias"),
    ("coarse 1", B * 257, 4608, 512, "plain"),
    ("coarse 2", B * 64, 4608, 512, "plain"),
    ("dec fine", B * NP, 512, 268, "gather"),
    ("dec 512-512", B * NP, 512, 512, "bn"),
    ("dec 512-256", B * NP, 256, 512, "bn"),
    ("dec 256-128", B * NP, 128, 256, "bn"),
    ("conv_0 last", B * NP, 128, 132, "layer"),
    ("conv_1 last", B * NP, 128, 128, "layer2"),
    ("conv_2 last", B * 257, 256, 256, "layer2"),
    ("conv_4 last", B * 64, 512, 512, "layer2_64"),
]
CONFIGS = (4, 5, 8)


def timeit(fns, reps=4, rounds=7):
    """interleaved rounds: every candidate once per round; median per candidate (first round dropped)"""
    ts = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / reps * 1e3)
    return {k: sorted(v[1:])[len(v[1:]) // 2] for k, v in ts.items()}


for name, M, N, K, fl in SHAPES:
    lda = (K + 3) // 4 * 4
    A = torch.randn(M, lda, device=dev)
    W = (torch.randn(N, lda, device=dev) / K ** 0.5).contiguous()
    if lda > K:
        W[:, K:] = 0
    Ws = ops.split_w(W)
    Wp = ops.planes_w(W[:, :K].contiguous())
    Ap = ops.planes_split(A, K=K)
    kw = dict(M=M, N=N, K=K, lda=lda, ldw=lda, ldc=N, w_split=Ws)
    n_obj = {"layer2_64": 64}.get(fl, NP if M == B * NP else 257)
    if fl != "plain":
        kw["bias"] = torch.randn(N, device=dev)
    if fl in ("bn", "gather", "layer", "layer2", "layer2_64"):
        kw.update(scale=torch.rand(N, device=dev) + 0.5, shift=torch.randn(N, device=dev), act=1)
    if fl in ("layer", "layer2", "layer2_64", "gather"):
        kw.update(rowbias=torch.randn(M // n_obj, N, device=dev), rows_per_obj=n_obj)
    if fl in ("layer", "layer2", "layer2_64"):
        kw.update(res1=torch.randn(M, N, device=dev), ldr1=N)
    if fl in ("layer2", "layer2_64"):
        kw.update(res2=torch.randn(M, N, device=dev), ldr2=N)
    if fl == "gather":
        P1 = torch.randn(B * 257, 4608, device=dev)
        P2 = torch.randn(B * 64, 4608, device=dev)
        i1 = torch.sort(torch.randint(0, 257, (B, NP), device=dev), dim=1)[0] + 257 * torch.arange(B, device=dev)[:, None]
        i2 = torch.sort(torch.randint(0, 64, (B, NP), device=dev), dim=1)[0] + 64 * torch.arange(B, device=dev)[:, None]
        kw.update(gather1=(P1[:, 4096:], 4608, i1.int().contiguous().view(-1)), gather2=(P2[:, 4096:], 4608, i2.int().contiguous().view(-1)))
    C0 = torch.empty(M, N, device=dev)
    ops.gemm(A, W, C0, **kw)
    fns = {"old": lambda: ops.gemm(A, W, C0, **kw)}
    Cs = {}
    bad = []
    for cfg in CONFIGS:
        Cs[cfg] = torch.zeros(M, N, device=dev)
        try:
            ops.gemm(A, W, Cs[cfg], a_planes=Ap, w_planes=Wp, pp_config=cfg, **kw)
        except Exception as e:        # a refused configuration (epilogue preconditions)
            say("   config %d refused: %s" % (cfg, e))
            del Cs[cfg]
            continue
        torch.cuda.synchronize()
        if not torch.equal(Cs[cfg], C0):
            bad.append((cfg, float((Cs[cfg] - C0).abs().max())))
        fns["pp%d" % cfg] = (lambda c: (lambda: ops.gemm(A, W, Cs[c], a_planes=Ap, w_planes=Wp, pp_config=c, **kw)))(cfg)
    # result planes: the consumer's operand written by the producer's epilogue == the stand-alone split of the fp32 result
    Cp = ops.Planes(M, N, dev)
    Cx = torch.zeros(M, N, device=dev)
    ops.gemm(A, W, Cx, a_planes=Ap, w_planes=Wp, c_planes=Cp, **kw)
    Cref = ops.planes_split(C0)
    torch.cuda.synchronize()
    nb = (M + 31) // 32
    valid_rows = torch.arange(nb * 32, device=dev).view(nb, 1, 1, 1, 32, 1) < M
    pl_ok = bool(((Cp.buf.view(nb, Cp.kt, 2, 2, 32, 16) == Cref.buf.view(nb, Cp.kt, 2, 2, 32, 16)) | ~valid_rows).all())
    am_ok = torch.equal(Cp.amax, Cref.amax)
    if fl.startswith("layer"):      # fp32 operand, result planes: the small-tile split kernel's planes-writing instance
        Cp2 = ops.Planes(M, N, dev)
        Cy = torch.zeros(M, N, device=dev)
        ops.gemm(A, W, Cy, c_planes=Cp2, **kw)
        torch.cuda.synchronize()
        ok2 = bool(((Cp2.buf.view(nb, Cp.kt, 2, 2, 32, 16) == Cref.buf.view(nb, Cp.kt, 2, 2, 32, 16)) | ~valid_rows).all())
        say("   old kernel + planes out: C equal %s, planes equal %s, amax equal %s" % (torch.equal(Cy, C0), ok2, torch.equal(Cp2.amax, Cref.amax)))
        fns["old+planes"] = lambda: ops.gemm(A, W, Cy, c_planes=Cp2, **kw)
    fns["pp0+planes"] = lambda: ops.gemm(A, W, Cx, a_planes=Ap, w_planes=Wp, c_planes=Cp, **kw)
    fns["pp0"] = lambda: ops.gemm(A, W, Cx, a_planes=Ap, w_planes=Wp, **kw)
    t = timeit(fns)
    gf = 2.0 * M * N * K / 1e3
    say("%-12s M=%6d N=%5d K=%4d %-9s | " % (name, M, N, K, fl) + "  ".join("%s %6.1f us (%3.0f TF)" % (k, v, gf / v / 1e3) for k, v in t.items()))
    say("   bit-identical: %s   planes out == split(C): %s   amax: %s" % ("yes" if not bad else "NO %s" % bad, pl_ok, am_ok))
