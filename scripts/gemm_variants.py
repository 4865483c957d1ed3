"""Development micro-benchmark (GPU box): interleaved A/B timing of the fp32 MFMA GEMM main-loop variants
in tg-pose_amd/csrc/gemm_variants.hip, plus the sustained shader clock during them."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgpose_amd import _lib

from _dev import use_dev_lib
lib = use_dev_lib()
fn = lib.tgp_debug_gemm_variant
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
M, N, K = 32768, 4096, 1280
LD = int(os.environ.get("VAR_LD", "1292"))
A = torch.randn(M, LD, device=dev)
W = torch.randn(N, LD, device=dev) / K ** 0.5
C = torch.empty(M, N, device=dev)
ref = None
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "128x128 dbuf BK32 (production)", 1: "128x128 single BK32", 2: "128x128 dbuf BK16", 3: "256x128 512thr dbuf",
         4: "128x256 512thr dbuf", 5: "256x128 512thr single", 6: "128x128 single BK64", 7: "256x256 1024thr single",
         8: "128x128 single BK16", 9: "256x256 1024thr dbuf", 10: "256x256 512thr single (64x128 wave)",
         11: "256x256 1024thr BK16 dbuf", 12: "256x128 512thr BK16 dbuf", 13: "256x256 1024thr BK16 single",
         14: "128x128 1024thr BK16 dbuf (tail)", 15: "256x128 512thr BK16 single", 16: "128x256 512thr BK16 dbuf",
         17: "256x256 1024thr BK16 dbuf 32x128w", 18: "256x256 1024thr BK16 dbuf 128x32w", 19: "128x128 256thr BK16 dbuf",
         20: "256x128 1024thr BK16 dbuf 64x32w"}
if os.environ.get("VAR_ONLY"):
    names = {int(v): names[int(v)] for v in os.environ["VAR_ONLY"].split(",")}
ok = []
for v in names:
    C.zero_()
    rc = fn(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, LD, v, None, stream)
    torch.cuda.synchronize()
    if rc != 0:
        print("variant", v, "rc", rc); continue
    if ref is None:
        ref = (A[:512, :K].double() @ W[:, :K].double().t()).float()
    err = (C[:512] - ref).abs().max().item()
    print("variant %d %-32s max err vs fp64 (first 512 rows) %.2e" % (v, names[v], err))
    if err < 1e-3: ok.append(v)
times = {v: [] for v in ok}
for rnd in range(6):
    for v in ok:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, LD, v, None, stream)
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 3)
flops = 2.0 * M * N * K
for v in ok:
    t = sorted(times[v][1:])
    print("variant %d %-32s median %.3f ms  %.1f TF   min %.3f ms %.1f TF" % (v, names[v], t[len(t)//2], flops / t[len(t)//2] / 1e9, t[0], flops / t[0] / 1e9))
# clock: stamps from one variant after a sustained run
BEST = int(os.environ.get('VAR_BEST', '11'))
nb = (M // 128) * (N // 128)   # upper bound on the number of blocks of any variant
st = torch.zeros(nb * 4, dtype=torch.int64, device=dev)
for _ in range(20):
    fn(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, LD, BEST, None, stream)
fn(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, LD, BEST, st.data_ptr(), stream)
torch.cuda.synchronize()
s = st.view(nb, 4).cpu().double()
cyc, real = s[:, 1] - s[:, 0], s[:, 3] - s[:, 2]
clk = (cyc / real * 100.0)       # MHz: memrealtime ticks at 100 MHz
print("in-kernel clock during the production GEMM: median %.0f MHz (p10 %.0f, p90 %.0f); block duration median %.1f us"
      % (clk.median().item(), clk.quantile(0.1).item(), clk.quantile(0.9).item(), (real.median() / 100.0).item()))
