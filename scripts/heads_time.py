"""Times tgp_heads_fused (and tgp_conv_max_fused) alone at the benchmark's shape (B = 32, N = 1028: 771 workgroups) on random operands.

    python scripts/heads_time.py [--rounds 5] [--reps 20] [--points 1028] [--batch 32]
    python scripts/heads_time.py --stamps            # development library: per-wave cycle stamps of the heads kernel's sections
    python scripts/heads_time.py --knobs 0,1,2,4,7   # development library: timing-only builds (1 = no LDS-DMA after the prologue,
                                                     # 2 = no gathers of the coarse products, 4 = no fragment reads; results are garbage;
                                                     # heads kernel: 0, 1, 2, 4, 7; conv_5 kernel: 0, 1, 2, 4, 8 = no epilogue)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=1028)
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--knobs", default="")
    a = ap.parse_args()
    dev_lib = None
    if a.stamps or a.knobs:
        from _dev import use_dev_lib
        dev_lib = use_dev_lib()
    from tgpose_amd import ops
    dev = "cuda:0"
    B, N, K, heads = a.batch, a.points, 268, 3
    M = B * N
    gen = torch.Generator().manual_seed(1)
    fine = torch.randn(M, 272, generator=gen)
    fine[:, K:] = 0
    Wa = torch.randn((heads + 1) * 1024, 272, generator=gen) / K ** 0.5
    Wa[:, K:] = 0
    n1, n2 = B * (N // 4), B * (N // 16)
    P1, P2 = torch.randn(n1, 4096, generator=gen), torch.randn(n2, 4096, generator=gen)
    # sorted parents, as the engine's rows are (a 32-row wave tile meets few distinct coarse rows)
    idx2 = torch.sort(torch.randint(0, n2, (M,), generator=gen, dtype=torch.int32))[0]
    idx1 = torch.sort(torch.randint(0, n1, (M,), generator=gen, dtype=torch.int32))[0]
    bias, scale, shift = (torch.randn(4096, generator=gen) * 0.1, torch.rand(4096, generator=gen) + 0.5, torch.randn(4096, generator=gen) * 0.1)
    W2 = torch.randn(heads, 256, 1024, generator=gen) / 32.0
    b2, sc2, sh2 = (torch.randn(heads, 256, generator=gen) * 0.1, torch.rand(heads, 256, generator=gen) + 0.5, torch.randn(heads, 256, generator=gen) * 0.1)
    d = lambda t: t.contiguous().to(dev)
    fine_d = d(fine)
    was = ops.split_f16(d(Wa))
    pl = ops.planes_split(fine_d, K=K, kt=17)
    wap = ops.heads_planes_w(d(Wa)[1024:])
    wcp = ops.heads_planes_w(d(Wa)[:1024])
    w2p = ops.heads_pack_w2(d(W2), d(bias)[1024:], d(scale)[1024:], d(shift)[1024:])
    g = dict(P1=d(P1), P2=d(P2), idx1=d(idx1), idx2=d(idx2), bias=d(bias), scale=d(scale), shift=d(shift), b2=d(b2), sc2=d(sc2), sh2=d(sh2))

    def heads_():
        return ops.heads_fused(fine_d, K, wap, g["P1"][:, 1024:], g["idx1"], g["P2"][:, 1024:], g["idx2"], w2p, g["b2"], g["sc2"], g["sh2"],
                               B, N, fine_planes=pl)

    def conv5():
        return ops.conv_max_fused(fine_d, K, wcp, g["P1"], g["idx1"], g["P2"], g["idx2"], g["bias"][:1024], g["scale"][:1024],
                                  g["shift"][:1024], 0.2, B, N, fine_planes=pl)

    if a.stamps:
        st = torch.zeros(3 * ((M + 127) // 128) * 4 * 12, device=dev, dtype=torch.int64)
        for _ in range(3):
            heads_()
        assert dev_lib.tgp_debug_set_heads_stamps(ctypes_ptr(st)) == 0
        heads_()
        torch.cuda.synchronize()
        dev_lib.tgp_debug_set_heads_stamps(None)
        t = st.view(-1, 4, 12).cpu().double()
        names = ["prologue", "channel-block loop", "epilogue 2", "last iteration", "stage A (conv1 + epilogue 1)", "-", "stage B (conv2)",
                 "barrier", "total cycles", "total wall x10ns"]
        clk = (t[:, :, 8] / (t[:, :, 9] * 10e-3)).median().item()      # cycles per us
        print("workgroups %d; shader clock over a workgroup's life: %.0f MHz (median)" % (t.shape[0], clk))
        for i, n in enumerate(names):
            if n == "-":
                continue
            v = t[:, :, i].flatten()
            print("  %-30s median %9.0f   p10 %9.0f   p90 %9.0f   (%.1f us at that clock)"
                  % (n, v.median().item(), v.quantile(0.1).item(), v.quantile(0.9).item(), v.median().item() / clk))
        w0 = t[:, 0, 10]
        w0 = (w0 - w0.min()) / 100.0
        dur = t[:, 0, 9] / 100.0
        order = torch.argsort(w0)
        print("workgroup start times (us), sorted, every 32nd:", [round(w0[i].item(), 1) for i in order[::32]])
        print("kernel span from first entry to last exit: %.1f us" % ((w0 + dur).max().item()))
        return
    knobs = [int(k) for k in a.knobs.split(",")] if a.knobs else [0]
    for name, fn in (("heads_fused", heads_), ("conv_max_fused", conv5)):
        times = {k: [] for k in knobs}
        for rnd in range(a.rounds):
            for k in knobs:
                if dev_lib is not None:
                    dev_lib.tgp_debug_set_heads_knobs(k)
                keys, over = fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / a.reps * 1e3)
        for k in knobs:
            t = sorted(times[k])
            print("%-16s knobs=%d  median %7.1f us   min %7.1f   max %7.1f" % (name, k, t[len(t) // 2], t[0], t[-1]), flush=True)


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


if __name__ == "__main__":
    main()
