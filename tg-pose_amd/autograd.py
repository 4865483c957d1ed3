"""Training-mode forward WITH autograd: PoseNet9D.forward as a graph of torch.autograd.Functions whose forward and
backward both run on the HIP kernels (C ABI of include/tgpose.h), so that ``loss.backward()`` (trainer/RL_TDA.py:205-224)
produces the gradients of every parameter.

torch's autograd engine only sequences the backward and carries the glue that moves no arithmetic of its own
(concatenation of the per-level features, the residual / broadcast adds of an HS layer, the (B, <=8) head post-processing):
every GEMM, BatchNorm, graph convolution, neighbourhood max and scatter below is a HIP kernel.  The faster, fused
no-autograd paths of ``engine.py`` are untouched; this module trades fusion for differentiability (round 1: correctness).

Reference graph restated here: network/fs_net_repo/gcn3d.py:78-112,142-186,210-245, FaceRecon.py:39-86,112-117,139-167,
PoseR.py:26-39, PoseTs.py:31-45, PoseNet9D.py:33-91.
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import engine, ops

BN_EPS = engine.BN_EPS
FEAT_C, FEAT_LD = engine.FEAT_C, engine.FEAT_LD
S = 7


def _w2(conv):
    """(N, K) view of a kernel-size-1 Conv1d weight (N, K, 1).  A view, not `weight[:, :, 0]`: the select's backward is a zero fill
    plus a copy of every weight gradient (two launches and 2 x the weight's bytes per layer and step), a view's is free."""
    w = conv.weight
    return w.view(w.shape[0], w.shape[1])


def _pad4(t):
    """pad the last dim with zeros to a multiple of 4 (the GEMM kernels read 16-byte quads)"""
    r = (-t.shape[-1]) % 4
    return F.pad(t, (0, r)) if r else t


def _split_if_big(w, rows, grads=False):
    """operand split for the tile GEMM kernels.  Gradients span many orders of magnitude below 1, outside what the
    two-term fp16 split represents (fp16 bottoms out at 6e-8): the backward GEMMs use the three-term bf16 split, which
    keeps fp32's range."""
    if rows <= 32 or ops.GEMM_MODE == "fp32":
        return None
    return ops.split_bf16(w) if grads else ops.split_w(w, check=False)     # inside the (captured) step: no read-back


def _dx_ksplit(rows, Kx, N):
    """chunks of the reduction (over the layer's N outputs) for dx = dy W when its output has few tiles: a power of two that brings
    tiles x chunks to about half a round of the 256 workgroup slots and divides N into multiples of 16; 1 = no split"""
    tiles = ((rows + 255) // 256) * ((Kx + 255) // 256)
    if N < 2048 or tiles > 80:
        return 1
    Z = 1
    while Z * 2 * tiles <= 160 and N % (Z * 2 * 16) == 0 and N // (Z * 2) >= 256:
        Z *= 2
    return Z


def _linear_backward(x, W, dy, need_dx, need_dw, need_db, dx_accum=None, scale=None, dx_cols=None, dx_res=None):
    """dx = dy W (+ dx_accum, added by the GEMM's own epilogue), dW = dy^T x, db = colsum(dy) for y = x W^T + b; W contiguous (N, K).
    Both GEMMs of a large layer run on the fp16 split kernels with dy lifted into fp16's range by one power of two chosen on the
    device (ops.absmax_scale); small ones keep the bf16x3 / fp32 MFMA paths."""
    N, K = W.shape
    dy2 = dy.reshape(-1, N)
    x2 = x.reshape(-1, K)
    dx = dW = db = None
    rows = dy2.shape[0]
    Kx = K if dx_cols is None else dx_cols        # dx_cols: only the first dx_cols input columns need a gradient (the rest are data)
    dx_f16 = need_dx and ops.GEMM_MODE == "split16" and ops.TN_SPLIT and ops._routes_to_big_tile(rows, Kx, 1, True)
    dw_f16 = need_dw and ops.tn_split_ok(rows, N, K)
    dyc = dy2.contiguous()
    # (scale: absmax_scale(dy) when the producer of dy collected it on the way -- the BatchNorm backward's apply pass)
    sc = (scale if scale is not None else ops.absmax_scale(dyc)) if (dx_f16 or dw_f16) else None
    if need_dx:
        dyp = _pad4(dyc).contiguous()                       # the reduction dim of this GEMM is N
        Wd = W if dx_cols is None else W[:, :dx_cols]       # (a column slice: read through its row stride)
        if dx_f16:
            wt, wt_s = ops.transpose_both(Wd)               # (Kx, N16) fp32 and its fp16 planes, one pass over the weight
        else:
            wt, wt_s = _pad4(ops.transpose(Wd.contiguous())).contiguous(), None     # (Kx, N4)
        out = res = None
        if dx_accum is not None:                            # the running sum of the other consumers' gradients: read as the
            out = res = dx_accum.view(-1, Kx)               # epilogue's residual and overwritten in place
        if dx_res is not None:                              # dx = dy W + dx_res (a residual path's gradient), added by the epilogue
            res = dx_res.reshape(-1, Kx)
        shape = tuple(x.shape[:-1]) + (Kx,)
        Z = _dx_ksplit(rows, Kx, dyp.shape[1]) if (dx_f16 and dx_res is None) else 1
        if Z > 1:
            # few output tiles under a long reduction (the coarse levels' layers: 2048 or 8224 rows, N up to 4608): the reduction
            # is cut into Z chunks that run as the batch dimension of one launch; the partial products are added in order
            chunk = dyp.shape[1] // Z
            parts = torch.empty(Z, rows, Kx, device=dyp.device, dtype=torch.float32)
            ops.gemm(dyp, wt, parts, M=rows, N=Kx, K=chunk, lda=dyp.shape[1], ldw=wt.shape[1], ldc=Kx, batch=Z,
                     batch_strides=(chunk, 0, rows * Kx, 0, 0), w_split=wt_s, a_scale=sc, ksplit_chunk=chunk)
            dx = ops.sum_slabs(parts, sc[1:], out=out, accumulate=out is not None).view(shape)
        elif dx_f16:
            dx = ops.linear_rows(dyp, wt, w_split=wt_s, a_scale=sc, c_scale=sc[1:], out=out, res1=res).view(shape)
        else:
            dx = ops.linear_rows(dyp, wt, w_split=_split_if_big(wt, dyp.shape[0], grads=True), out=out, res1=res).view(shape)
    if need_dw:
        dW = ops.gemm_tn(dyc, x2, scale=sc if dw_f16 else None)
    if need_db:
        db = ops.colsum(dyc)
    return dx, dW, db


class _Linear(Function):
    """y = x W^T (+ b): x (..., K) rows, W (N, K).  dW = dy^T x (TN GEMM), dx = dy W (forward kernel on W^T), db = colsum."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = x.contiguous()
        Wc = W.contiguous()                                 # once: a column slice of a weight (conv2's halves) is a copy
        rows = x.numel() // x.shape[-1]
        y = ops.linear_rows(x, Wc, bias=b, w_split=_split_if_big(Wc, rows))
        ctx.save_for_backward(x, Wc)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        return _linear_backward(x, W, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2],
                                scale=getattr(dy, "_tgp_scale", None))


class _LinearEpi(Function):
    """y = x W^T (+ b) (+ rb broadcast over each object's points) (+ res) in ONE launch: the GEMM's epilogue adds the per-object row
    bias and the residual (the ORL block's `conv2(cat[g, global]) + g`, gcn3d.py:108-112, and `... + STE(...)`, :87-89 / :147-152,
    were a GEMM plus two element-wise passes forward and a gradient add backward).  res_is_x: the residual is x itself (then
    dx = dy W + dy, the second term through the dx GEMM's epilogue)."""

    @staticmethod
    def forward(ctx, x, W, b, rb, res, res_is_x):
        x = x.contiguous()
        Wc = W.contiguous()
        rows = x.numel() // x.shape[-1]
        r = x if res_is_x else res
        y = ops.linear_rows(x, Wc, bias=b, rowbias=None if rb is None else rb.contiguous(), rows_per_obj=x.shape[-2] if rb is not None else 0,
                            res1=None if r is None else r.contiguous(), w_split=_split_if_big(Wc, rows))
        ctx.save_for_backward(x, Wc)
        ctx.flags = (b is not None, rb is not None, res is not None, bool(res_is_x))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        has_b, has_rb, has_res, same = ctx.flags
        need = ctx.needs_input_grad
        dyc = dy.contiguous()
        dx, dW, db = _linear_backward(x, W, dyc, need[0], need[1], has_b and need[2], scale=getattr(dy, "_tgp_scale", None),
                                      dx_res=dyc if (same and need[0]) else None)
        drb = ops.colsum_objects(dyc) if (has_rb and need[3]) else None
        return dx, dW, db, drb, (dy if (has_res and need[4]) else None), None


class _FeatConsumers(Function):
    """The five layers that read the concat buffer `feat` (conv_5 of the PH predictor, the decoder's first conv, conv1 of the three
    heads; FaceRecon.py:76,112,  PoseR.py:26, PoseTs.py:31) as ONE autograd node: y_i = feat W_i^T (+ b_i) with W_i zero-padded to
    feat's row stride inside the node.  Forward: the same five GEMMs.  Backward: d feat is accumulated by the dx GEMMs' own
    epilogues (each reads the running sum as its residual and overwrites it), where five separate nodes made autograd add five
    170 MB tensors pairwise (0.38 ms per step) and pad / un-pad every weight through its own nodes."""

    @staticmethod
    def forward(ctx, feat, *wb):
        feat = feat.contiguous()
        rows = feat.numel() // feat.shape[-1]
        outs, saved, cols = [], [], []
        for i in range(0, len(wb), 2):
            W, b = wb[i], wb[i + 1]
            Wp = F.pad(W, (0, feat.shape[-1] - W.shape[1])).contiguous()
            outs.append(ops.linear_rows(feat, Wp, bias=b, w_split=_split_if_big(Wp, rows)))
            saved.append(Wp)
            cols.append(W.shape[1])
        ctx.save_for_backward(feat, *saved)
        ctx.cols, ctx.has_bias = cols, [wb[i + 1] is not None for i in range(0, len(wb), 2)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        feat, *Ws = ctx.saved_tensors
        need = ctx.needs_input_grad
        dfeat, grads = None, []
        for i, (g, Wp) in enumerate(zip(gs, Ws)):
            if g is None:
                grads += [None, None]
                continue
            dx, dW, db = _linear_backward(feat, Wp, g, need[0], need[1 + 2 * i], ctx.has_bias[i] and need[2 + 2 * i], dx_accum=dfeat)
            dfeat = dx if dx is not None else dfeat
            grads += [dW[:, : ctx.cols[i]] if dW is not None else None, db]
        return (dfeat,) + tuple(grads)


class _FeatConsumersFactored(Function):
    """The same five layers with the nearest-neighbour up-sampling factored out, as the eval forward runs them (engine.pack_factored):
    feat = [fm_0 | fm_1 | up_1(fm_2) | up_1(fm_3) | up_2(fm_4) | tail] with up() a row fetch (FaceRecon.py:70-75), so
        y_i = fine Wa_i^T + P1[near1][:, cols_i] + P2[near2][:, cols_i] + b_i,   P1 = [fm_2 | fm_3] Wb^T,  P2 = fm_4 Wc^T
    with fine = [fm_0 | fm_1 | tail] and Wa_i / Wb / Wc the matching column blocks of the W_i (Wb, Wc stacked over the five layers):
    the 1024 up-sampled columns are multiplied once per COARSE point (N/4 and N/16 of them).  131 instead of 392 GFLOP forward at
    B = 32, N = 1028, and the same ratio in both backward GEMMs:
        d fine += g_i Wa_i,  dWa_i = g_i^T fine,  d P1[r] = sum of g[i] over the children i of r (tgp_segsum_rows: child lists, no
        atomics),  d[fm_2 | fm_3] = dP1 Wb,  dWb = dP1^T [fm_2 | fm_3]   (level 2 alike)
    Inputs: fm01 = [fm_0 | fm_1] (B, N, 256), [fm_2 | fm_3], fm_4, tail (B, N, 16: one-hot | xyz | 0, no gradient); near1 / near2:
    (B, N) int32 GLOBAL coarse rows (b * N1 + nearest); lists1 / lists2: ops.child_lists of the two levels."""

    @staticmethod
    def forward(ctx, fm01, fm23, fm4, tail, near1, near2, lists1, lists2, *wb):
        # fine = [fm_0 | fm_1 | tail]: built here, so that the node's differentiable input is the 256 feature columns only -- the
        # tail (one-hot, xyz, padding) is data, and d fine over 272 columns made every dx GEMM 2.1 column tiles wide (190 us each)
        fine = torch.cat([fm01, tail], 2)
        fm23, fm4 = fm23.contiguous(), fm4.contiguous()
        B, N, ldf = fine.shape
        N1, N2 = fm23.shape[1], fm4.shape[1]
        M = B * N
        Ws, bs = wb[0::2], wb[1::2]
        Wb = torch.cat([W[:, 256:768] for W in Ws], 0).contiguous()
        Wc = torch.cat([W[:, 768:1280] for W in Ws], 0).contiguous()
        P1 = ops.linear_rows(fm23.view(B * N1, -1), Wb, w_split=_split_if_big(Wb, B * N1))
        P2 = ops.linear_rows(fm4.view(B * N2, -1), Wc, w_split=_split_if_big(Wc, B * N2))
        ld = P1.shape[1]
        outs, Was, tails = [], [], []
        off = 0
        for W, b in zip(Ws, bs):
            n, kt = W.shape[0], W.shape[1] - 1280
            Wa = F.pad(torch.cat([W[:, :256], W[:, 1280:]], 1), (0, ldf - 256 - kt)).contiguous()
            y = torch.empty(B, N, n, device=fine.device, dtype=torch.float32)
            ops.gemm(fine, Wa, y, M=M, N=n, K=ldf, lda=ldf, ldw=ldf, ldc=n, bias=b, rows_per_obj=N, w_split=_split_if_big(Wa, M),
                     gather1=(P1[:, off:], ld, near1), gather2=(P2[:, off:], ld, near2), flops_ref=2.0 * M * n * W.shape[1])
            outs.append(y)
            Was.append(Wa)
            tails.append(kt)
            off += n
        (ptr1, idx1), (ptr2, idx2) = lists1, lists2
        ctx.save_for_backward(fine, fm23, fm4, ptr1, idx1, ptr2, idx2, Wb, Wc, *Was)
        ctx.tails, ctx.has_bias, ctx.nfeat = tails, [b is not None for b in bs], fm01.shape[2]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        fine, fm23, fm4, ptr1, idx1, ptr2, idx2, Wb, Wc, *Was = ctx.saved_tensors
        need = ctx.needs_input_grad
        M = fine.shape[0] * fine.shape[1]
        ld = Wb.shape[0]
        dP1 = torch.empty(ptr1.numel() - 1, ld, device=fine.device, dtype=torch.float32)
        dP2 = torch.empty(ptr2.numel() - 1, ld, device=fine.device, dtype=torch.float32)
        dfine, part, off = None, [], 0
        need_w = any(need[8 + 2 * i] for i in range(len(Was)))
        for i, (g, Wa) in enumerate(zip(gs, Was)):
            n = Wa.shape[0]
            if g is None:
                dP1[:, off:off + n] = 0
                dP2[:, off:off + n] = 0
                part.append((None, None))
            else:
                gc = g.reshape(M, n).contiguous()
                ops.segsum_rows(gc, ptr1, idx1, out=dP1[:, off:off + n])
                ops.segsum_rows(gc, ptr2, idx2, out=dP2[:, off:off + n])
                dx, dWa, db = _linear_backward(fine, Wa, gc, need[0], need[8 + 2 * i], ctx.has_bias[i] and need[9 + 2 * i], dx_accum=dfine,
                                               scale=getattr(g, "_tgp_scale", None), dx_cols=ctx.nfeat)
                dfine = dx if dx is not None else dfine
                part.append((dWa, db))
            off += n
        dfm23, dWb, _ = _linear_backward(fm23, Wb, dP1, need[1], need_w, False)
        dfm4, dWc, _ = _linear_backward(fm4, Wc, dP2, need[2], need_w, False)
        grads, off = [], 0
        for (dWa, db), Wa, kt in zip(part, Was, ctx.tails):
            n = Wa.shape[0]
            dW = None
            if dWa is not None:
                dW = torch.cat([dWa[:, :256], dWb[off:off + n], dWc[off:off + n], dWa[:, 256:256 + kt]], 1)
            grads += [dW, db]
            off += n
        return (dfine, dfm23, dfm4, None, None, None, None, None) + tuple(grads)


FACTORED = os.environ.get("TGP_TRAIN_FACTORED", "1") != "0"     # the layers over the concat buffer factored over the up-sampling


def feat_consumers_factored(parts, layers):
    """parts: the encoder's (fm01, fm23, fm4, tail16, near1, near2, lists1, lists2); layers as feat_consumers (weights UNPADDED: (N, 1286 or 1289))"""
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _FeatConsumersFactored.apply(*parts, *flat)


def feat_consumers(feat, layers):
    """layers: [(weight (N, K <= row stride of feat), bias or None), ...] -> one output per layer"""
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _FeatConsumers.apply(feat, *flat)


def linear(x, W, b=None):
    return _Linear.apply(x, W, b)


class _BNAct(Function):
    """act(BatchNorm1d_train(x)) over rows (..., C); updates the module's running statistics like nn.BatchNorm1d."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, act, slope):
        x = x.contiguous()
        y = torch.empty_like(x)
        _, mean, var = ops.bn_train(x, gamma, beta, BN_EPS, act, slope, out=y, running=_running(bn))
        ctx.save_for_backward(x, mean, var, gamma, beta)
        ctx.act, ctx.slope = act, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, var, gamma, beta = ctx.saved_tensors
        # (large layers: the apply pass also collects max |dx|, which the backward of the linear layer in front needs for its scale)
        big = x.shape[-1] >= 128 and x.numel() // x.shape[-1] >= 2048
        dx, dg, db = ops.bn_bwd(dy.contiguous(), x, mean, var, gamma, beta, BN_EPS, ctx.act, ctx.slope, dx=torch.empty_like(x),
                                want_scale=big)
        return dx, dg, db, None, None, None


class _BNActPool(Function):
    """max over each object's points of act(BatchNorm1d_train(x)): x (B, n, C) -> (B, C)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, act, slope):
        x = x.contiguous()
        B, n, C = x.shape
        keys = torch.zeros(B, C, device=x.device, dtype=torch.int32)
        _, mean, var = ops.bn_train(x, gamma, beta, BN_EPS, act, slope, want_out=False, colmax_keys=keys, rows_per_obj=n,
                                    running=_running(bn))
        pooled, arg = ops.colmax_arg(x, B, n, bn=(mean, var, gamma, beta), act=act, slope=slope)
        if TAPS is not None:
            _tap_seq("pool", (arg.detach().cpu() % n, pooled.detach().cpu()))
        ctx.save_for_backward(x, mean, var, gamma, beta, arg)
        ctx.act, ctx.slope = act, slope
        return pooled

    @staticmethod
    def backward(ctx, dpool):
        x, mean, var, gamma, beta, arg = ctx.saved_tensors
        B, n, C = x.shape
        dx, dg, db = ops.bn_bwd_pooled(dpool.contiguous(), arg, x, n, mean, var, gamma, beta, BN_EPS, ctx.act, ctx.slope)
        return dx.view(B, n, C), dg, db, None, None, None


def _running(bn):
    """the module's buffers for ops.bn_train(running=...): moved by the statistics kernel (was: four torch launches per module).
    momentum=None (torch's cumulative moving average, 1 / num_batches_tracked) is not implemented by that kernel and is refused
    rather than silently leaving the running statistics untouched; the reference's modules all use the default 0.1."""
    if bn is None or not bn.track_running_stats:
        return None
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative average) is not supported by the training path")
    return bn.running_mean, bn.running_var, bn.momentum, bn.num_batches_tracked


# tests (tests/test_gpu_parity.py::test_backward_full_network_with_forced_decisions): a dict that receives, on the host, the layer outputs
# (by name), every activation's output ("act.i", call order) and every pooled layer's winners and values ("pool.i") of a training
# forward -- the decisions of this run, which the CPU oracle then takes over (oracle.posenet_ref.posenet_forward(force=...)).
TAPS = None


def _tap(name, t):
    if TAPS is not None:
        TAPS[name] = t.detach().cpu()
    return t


def _tap_seq(kind, value):
    if TAPS is not None:
        i = TAPS["_n_" + kind] = TAPS.get("_n_" + kind, -1) + 1
        TAPS["%s.%d" % (kind, i)] = value


def bn_act(x, bn, act=1, slope=0.0):
    y = _BNAct.apply(x, bn.weight, bn.bias, bn, act, slope)
    if TAPS is not None:
        _tap_seq("act", y.detach().cpu())
    return y


def bn_act_pool(x, bn, act=1, slope=0.0):
    return _BNActPool.apply(x, bn.weight, bn.bias, bn, act, slope)


class _GConvSurface(Function):
    @staticmethod
    def forward(ctx, xyz, idx, sdn, C):
        g = ops.gconv_surface(xyz, idx, sdn.contiguous(), S, C)
        ctx.save_for_backward(xyz, idx, sdn)
        ctx.C = C
        return g

    @staticmethod
    def backward(ctx, dg):
        xyz, idx, sdn = ctx.saved_tensors
        return None, None, ops.gconv_surface_bwd(xyz, idx, sdn.contiguous(), dg.contiguous(), S, ctx.C), None


# The backward of the graph layers without float atomics (csrc/graph_bwd.hip): reverse neighbour lists + two dense passes.  0: the
# first version's scatter with hardware atomics (csrc/gconv_bwd.hip), also the fallback for shapes the gather kernels do not take.
SCATTER_FREE = os.environ.get("TGP_SCATTER_FREE", "1") != "0"


class _GConvHS(Function):
    """rev: ops.reverse_graph(idx, n) or None (-> the atomic scatter)"""

    @staticmethod
    def forward(ctx, xyz, idx, proj, sdn, C, rev=None):
        proj = proj.contiguous()
        sdn_c = sdn.contiguous()
        ctx.slots = None
        if rev is not None and ops.gconv_gather_ok(C, idx.shape[2], proj, sdn_c):
            # the forward kernel that also records every maximum's slot: the backward then starts from them (no second walk over
            # the k neighbour rows); the slots are consumed by the one backward pass a training step runs
            g, ctx.slots = ops.gconv_hs_slots(xyz, idx, proj, sdn_c, S, C)
        else:
            g = ops.gconv_hs(xyz, idx, proj, sdn_c, S, C)
        ctx.save_for_backward(xyz, idx, proj, sdn)
        ctx.C, ctx.rev = C, rev
        return g

    @staticmethod
    def backward(ctx, dg):
        xyz, idx, proj, sdn = ctx.saved_tensors
        dg = dg.contiguous()
        if ctx.rev is not None and ops.gconv_gather_ok(ctx.C, idx.shape[2], proj, dg):
            slots, ctx.slots = ctx.slots, None            # (a second backward over the same graph recomputes them)
            dproj, dsdn = ops.gconv_hs_bwd_gather(xyz, idx, ctx.rev, proj, sdn.contiguous(), dg, S, ctx.C, slots=slots)
        else:
            dproj, dsdn = ops.gconv_hs_bwd(xyz, idx, proj, sdn.contiguous(), dg.contiguous(), S, ctx.C)
        return None, None, dproj, dsdn, None, None


class _NbrMaxMean(Function):
    """get_ORL_global without the repeat: mean over points of the max over each point's neighbours -> (B, C)"""

    @staticmethod
    def forward(ctx, g, idx, rev=None):
        g = g.contiguous()
        ctx.save_for_backward(g, idx)
        ctx.rev = rev
        return ops.orl_global(g, idx)

    @staticmethod
    def backward(ctx, dglob):
        g, idx = ctx.saved_tensors
        dglob = dglob.contiguous()
        if ctx.rev is not None and ops.nbrmax_gather_ok(g.shape[2], g, dglob):
            return ops.nbrmax_bwd_gather(g, idx, ctx.rev, dglob, per_object=True, scale=1.0 / g.shape[1]), None, None
        return ops.nbrmax_bwd(g, idx, dglob, per_object=True, scale=1.0 / g.shape[1]), None, None


class _PoolMax(Function):
    """Pool_layer's feature half: max over the kpool nearest neighbours at the sampled points"""

    @staticmethod
    def forward(ctx, xyz, fm, idx, sample, kpool=4):
        fm = fm.contiguous()
        v, f = ops.pool(xyz, fm, idx, sample, kpool=kpool)
        idx_s = idx[:, sample.long(), :kpool].contiguous()
        ctx.save_for_backward(fm, idx_s)
        ctx.rev = ops.reverse_graph(idx_s, fm.shape[1]) if SCATTER_FREE and fm.requires_grad else None
        ctx.mark_non_differentiable(v)
        return v, f

    @staticmethod
    def backward(ctx, _dv, df):
        fm, idx_s = ctx.saved_tensors
        df = df.contiguous()
        if ctx.rev is not None and ops.nbrmax_gather_ok(fm.shape[2], fm, df):
            return None, ops.nbrmax_bwd_gather(fm, idx_s, ctx.rev, df), None, None, None
        return None, ops.nbrmax_bwd(fm, idx_s, df), None, None, None


class _GatherRows(Function):
    """nearest up-sampling out[b, i] = fm[b, near[b, i]] (FaceRecon.py:70-75).  lists: ops.child_lists(near, n_src) -- the backward
    is then a segment sum per source row (no atomics); None: the atomic scatter."""

    @staticmethod
    def forward(ctx, fm, near, lists=None):
        fm = fm.contiguous()
        B, n_src, C = fm.shape
        out = torch.empty(B, near.shape[1], C, device=fm.device, dtype=torch.float32)
        ops.gather_rows(fm, near, out)
        ctx.save_for_backward(near)
        ctx.n_src, ctx.lists = n_src, lists
        return out

    @staticmethod
    def backward(ctx, dy):
        (near,) = ctx.saved_tensors
        B, n, C = dy.shape
        if ctx.lists is not None and C % 4 == 0:
            # (a column slice of the 1286-wide gradient of the concat buffer has 8-byte-aligned rows: tgp_segsum_rows' 8-byte form)
            if not (dy.stride(2) == 1 and dy.stride(0) == n * dy.stride(1) and dy.stride(1) % 2 == 0 and dy.data_ptr() % 8 == 0):
                dy = dy.contiguous()
            rows = dy.view(B * n, C) if dy.is_contiguous() else dy.as_strided((B * n, C), (dy.stride(1), 1))
            return ops.segsum_rows(rows, ctx.lists[0], ctx.lists[1]).view(B, ctx.n_src, C), None, None
        return ops.gather_rows_bwd(dy.contiguous(), near, ctx.n_src), None, None


class _AddRowBias(Function):
    """x (B, n, C) + rb (B, C) broadcast over each object's points.  The backward's sum over the points is tgp_colsum_objects,
    not torch's sum: a multi-block torch reduction inside the captured step did not replay reliably."""

    @staticmethod
    def forward(ctx, x, rb):
        return x + rb.unsqueeze(1)

    @staticmethod
    def backward(ctx, dy):
        return dy, ops.colsum_objects(dy.contiguous())


def add_row_bias(x, rb):
    return _AddRowBias.apply(x, rb)


class _HeadPost(Function):
    """PoseNet9D.py:57-66 on the heads' raw outputs -- axis / (|axis| + 1e-6), sigmoid of the confidences, T = ts[:3] + mean,
    s = ts[3:] -- as one launch each way (tgp_head_post, the eval forward's kernel, and tgp_head_post_bwd)"""

    @staticmethod
    def forward(ctx, green, red, ts, mean):
        green, red, ts = green.contiguous(), red.contiguous(), ts.contiguous()
        ctx.save_for_backward(green, red)
        return ops.head_post(green, red, ts, mean)

    @staticmethod
    def backward(ctx, *grads):
        green, red = ctx.saved_tensors
        dg, dr, dt = ops.head_post_bwd(green, red, grads)
        return dg, dr, dt, None


class _NormalizeDirs(Function):
    """F.normalize(directions, dim=0) (gcn3d.py:93, :160) as one launch each way (tgp_normalize_dirs / _bwd)"""

    @staticmethod
    def forward(ctx, d):
        d = d.contiguous()
        ctx.save_for_backward(d)
        return ops.normalize_dirs(d)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return ops.normalize_dirs_bwd(d, g)


class _SplitCols(Function):
    """w (N, K) -> (w[:, :c], w[:, c:]) as contiguous tensors; the backward is one concatenation (two column slices under autograd
    cost two zero fills, two copies and an add per layer and step)"""

    @staticmethod
    def forward(ctx, w, c):
        return w[:, :c].contiguous(), w[:, c:].contiguous()

    @staticmethod
    def backward(ctx, ga, gb):
        return torch.cat([ga, gb], 1), None


class _ColMax(Function):
    """feat_global = feat.max over an object's points (PoseNet9D.py:50): x (B, n, C) rows (row stride >= C) -> (B, C); first row
    wins ties and NaN propagates, as torch.max; the gradient lands on the winning row"""

    @staticmethod
    def forward(ctx, x):
        B, n, C = x.shape
        out, arg = ops.colmax_arg(x, B, n)
        ctx.save_for_backward(arg)
        ctx.shape = (B, n, C)
        return out

    @staticmethod
    def backward(ctx, dpool):
        (arg,) = ctx.saved_tensors
        B, n, C = ctx.shape
        return ops.colmax_bwd(dpool.contiguous(), arg, n).view(B, n, C)


def colmax(x):
    return _ColMax.apply(x)


# ------------------------------------------------------------------------------------------------------------------
def _reverse(graphs, idx, n_src):
    """reverse lists of a graph for the scatter-free backward: from the forward's graph source when it keeps them (one per distinct
    graph), else computed here (the stand-alone layer modules of the seam)"""
    rev = getattr(graphs, "rev", None)
    if rev is not None:
        return rev(idx, n_src)
    if not (SCATTER_FREE and torch.is_grad_enabled()):
        return None
    with torch.no_grad():
        return ops.reverse_graph(idx, n_src)


def _orl(layer, g, idx_orl, rev=None):
    """ORL_forward (gcn3d.py:108-112,182-186): conv2(cat[g, global]) + g, with the concatenation split into the two
    halves of conv2's weight (the global half is one row per object)."""
    C = g.shape[-1]
    w_pt, w_glob = _SplitCols.apply(_w2(layer.conv2), C)
    glob = _NbrMaxMean.apply(g, idx_orl, rev)                             # (B, C)
    return _LinearEpi.apply(g, w_pt, None, linear(glob, w_glob), None, True)


def _surface(layer, xyz, graphs, kmax):
    C = layer.kernel_num
    sdn = _NormalizeDirs.apply(layer.directions)
    pre = getattr(getattr(graphs, "g", None), "prefix", "")
    g = _tap(pre + "conv_0.g", _GConvSurface.apply(xyz, graphs("conv_0.rf", 0, xyz, kmax), sdn, C))
    idx_orl = graphs("conv_0.orl_xyz", 0, xyz, kmax)
    out = _orl(layer, g, idx_orl, _reverse(graphs, idx_orl, xyz.shape[1]))
    return _tap(pre + "conv_0.out", _LinearEpi.apply(_pad4(xyz), _pad4(_w2(layer.STE_layer)), None, None, out, False))   # STE(xyz) + out


def _hs(layer, name, xyz, fm, graphs, level, k):
    C = layer.out_channel
    sdn = _NormalizeDirs.apply(layer.directions)
    idx_rf = graphs(name + ".rf", None, fm, k)
    pre = getattr(getattr(graphs, "g", None), "prefix", "")
    proj = _tap(pre + name + ".proj", linear(fm, layer.weights.t(), layer.bias))     # (B, n, 8C) = [centre | support]
    g = _tap(pre + name + ".g", _GConvHS.apply(xyz, idx_rf, proj, sdn, C, _reverse(graphs, idx_rf, xyz.shape[1])))
    idx_orl = graphs(name + ".orl_xyz", level, xyz, k)
    out = _orl(layer, g, idx_orl, _reverse(graphs, idx_orl, xyz.shape[1]))
    return _tap(pre + name + ".out", _LinearEpi.apply(fm, _w2(layer.STE_layer), None, None, out, False))      # STE(fm) + out


class _GraphSource(object):
    """neighbour lists of one forward (computed by the HIP kNN kernels on detached tensors, shared per level like
    engine.Graphs; tests may inject the reference's own lists)"""

    def __init__(self, device, inject, record, prefix):
        self.g = engine.Graphs(device, inject, record, prefix)
        self.xyz = {}
        self._rev = {}

    def rev(self, idx, n_src):
        """reverse lists of a graph of this forward for the scatter-free backward (one per distinct graph: the xyz graphs are shared
        by the layers of a level); None when gradients are off, the switch is off or the shape is unsupported"""
        if not (SCATTER_FREE and torch.is_grad_enabled()):
            return None
        key = (idx.data_ptr(), tuple(idx.shape), n_src)       # every graph of the forward is alive in self.g: pointers are unique
        if key not in self._rev:
            with torch.no_grad():
                self._rev[key] = (idx, ops.reverse_graph(idx, n_src))     # (idx held: its pointer cannot be handed out again)
        return self._rev[key][1]

    def __call__(self, name, level, x, k):
        def compute():
            with torch.no_grad():
                if level is None:
                    return ops.knn_feat(x.detach().contiguous(), k)
                if level not in self.xyz:
                    self.xyz[level] = ops.knn_xyz(x.detach().contiguous(), k)
                return self.xyz[level]
        return self.g.get(name, compute)


def encoder(enc, xyz, obj_id, sample_idx, graphs, kmax=20, n_cls=6):
    """Face_Enc.forward (FaceRecon.py:39-86) -> feat, and the operands of the factored layers over it.  FACTORED off: feat is
    (B, N, FEAT_LD) = [fm_0..fm_4 | one-hot | xyz | 0 0 0] (the layers' GEMM operand) and the operands are None; on: feat is
    (B, N, FEAT_C) = [fm_0..fm_4 | one-hot]."""
    B, N, _ = xyz.shape
    dev = xyz.device
    s1 = sample_idx[0].to(device=dev, dtype=torch.int32)
    s2 = sample_idx[1].to(device=dev, dtype=torch.int32)
    fm0 = torch.relu(_surface(enc.conv_0, xyz, graphs, kmax))
    if TAPS is not None:
        _tap_seq("act", fm0.detach().cpu())
    fm1 = bn_act(_hs(enc.conv_1, "conv_1", xyz, fm0, graphs, 0, kmax), enc.bn1)
    v1, fp1 = _PoolMax.apply(xyz, fm1, graphs("pool_1.xyz", 0, xyz, kmax), s1)
    k1 = min(kmax, v1.shape[1] // 8)
    fm2 = bn_act(_hs(enc.conv_2, "conv_2", v1, fp1, graphs, 1, k1), enc.bn2)
    fm3 = bn_act(_hs(enc.conv_3, "conv_3", v1, fm2, graphs, 1, k1), enc.bn3)
    v2, fp2 = _PoolMax.apply(v1, fm3, graphs("pool_2.xyz", 1, v1, k1), s2)
    k2 = min(kmax, v2.shape[1] // 8)
    fm4 = _hs(enc.conv_4, "conv_4", v2, fp2, graphs, 2, k2)
    with torch.no_grad():
        near1 = graphs.g.get("up_1", lambda: ops.nn1(xyz, v1)).view(B, N)
        near2 = graphs.g.get("up_2", lambda: ops.nn1(xyz, v2)).view(B, N)
        one_hot = torch.zeros(B, n_cls, device=dev).scatter_(1, obj_id.view(-1, 1).long(), 1)
        tail = torch.cat([one_hot.unsqueeze(1).expand(B, N, n_cls), xyz, torch.zeros(B, N, FEAT_LD - FEAT_C - 3, device=dev)], 2)
        base = torch.arange(B, device=dev, dtype=torch.int32).view(B, 1)
        near1g, near2g = near1 + base * v1.shape[1], near2 + base * v2.shape[1]
        # children of every coarse point: the backward of the up-sampling (here and in the factored layers) is a segment sum
        lists1 = ops.child_lists(near1, v1.shape[1]) if SCATTER_FREE or FACTORED else None
        lists2 = ops.child_lists(near2, v2.shape[1]) if SCATTER_FREE or FACTORED else None
    up1, up2 = (lists1, lists2) if SCATTER_FREE else (None, None)
    ups = [_GatherRows.apply(fm2, near1, up1), _GatherRows.apply(fm3, near1, up1), _GatherRows.apply(fm4, near2, up2)]
    if not FACTORED:
        return torch.cat([fm0, fm1] + ups + [tail], dim=2), None
    # factored: no GEMM reads the concat buffer, so it carries exactly the FEAT_C columns the outputs `feat` / `feat_global` have (no
    # xyz / padding columns: a column slice under autograd costs a zero fill and a copy of the whole (B, N, 1292) gradient)
    feat = torch.cat([fm0, fm1] + ups + [tail[:, :, :n_cls]], dim=2)
    # the operands of the factored form of the layers over feat (_FeatConsumersFactored): the columns that differ from point to
    # point, and the two coarse levels
    tail16 = F.pad(tail, (0, engine.FINE_LD - 256 - tail.shape[2]))
    return feat, (torch.cat([fm0, fm1], dim=2), torch.cat([fm2, fm3], dim=2), fm4, tail16, near1g, near2g, lists1, lists2)


def _w_feat(conv, cols=FEAT_C):
    """weight of a Conv1d that reads the concat buffer, zero-padded to its row stride (the xyz columns included for Pose_Ts)"""
    w = _w2(conv)
    return F.pad(w, (0, FEAT_LD - w.shape[1]))


def ph_predictor(ph, feat, x=None):
    """PH_Predictor.forward (FaceRecon.py:139-167) -> back = pi1_1 + pi2_1 (B, FEAT_LD-padded), h1, h2.  x: conv_5's raw output
    when the caller computed the layers over `feat` as one node (feat_consumers)"""
    if x is None:
        x = linear(feat, _w_feat(ph.conv_5[0]))
    g = bn_act_pool(x, ph.conv_5[1], act=1, slope=0.2)                     # (B, 1024)
    g = torch.cat((g, g), 1)
    fa = bn_act(linear(g, ph.linear1.weight), ph.bn5, act=1, slope=0.2)
    fa = ph.dp1(fa)
    pi1 = linear(fa, ph.linear2.weight, ph.linear2.bias)
    pi2 = linear(fa, ph.linear3.weight, ph.linear3.bias)
    back = linear(_pad4(pi1), _pad4(ph.linear4.weight), ph.linear4.bias) + linear(_pad4(pi2), _pad4(ph.linear5.weight), ph.linear5.bias)
    return back, torch.sigmoid(pi1), torch.sigmoid(pi2)


def decoder(dec, feat, back, x=None):
    """Face_Dec.forward on feat + back (FaceRecon.py:112-117,165): conv(feat + back) = conv(feat) + W back per object"""
    blk = dec.conv1d_block
    w0 = _w_feat(blk[0])
    if x is None:
        x = linear(feat, w0, blk[0].bias)
    if back is not None:
        x = add_row_bias(x, linear(F.pad(back, (0, FEAT_LD - back.shape[1])), w0))
    x = bn_act(x, blk[1])
    x = bn_act(linear(x, _w2(blk[3]), blk[3].bias), blk[4])
    x = bn_act(linear(x, _w2(blk[6]), blk[6].bias), blk[7])
    x = bn_act(linear(x, _w2(dec.recon_head[0]), dec.recon_head[0].bias), dec.recon_head[1])
    return linear(x, _w2(dec.recon_head[3]), dec.recon_head[3].bias)


def point_head(hd, feat, x=None):
    """Rot_green / Rot_red / Pose_Ts (PoseR.py:26-39, PoseTs.py:31-45) on the concat buffer -> (B, out)"""
    x = bn_act(linear(feat, _w_feat(hd.conv1), hd.conv1.bias) if x is None else x, hd.bn1)
    x = bn_act_pool(linear(x, _w2(hd.conv2), hd.conv2.bias), hd.bn2)
    x = bn_act(linear(x, _w2(hd.conv3), hd.conv3.bias), hd.bn3)
    x = hd.drop1(x)
    return linear(x, _w2(hd.conv4), hd.conv4.bias)


def proj_global(enc, feat):
    """feat_global with enable_proj=True (FaceRecon.py:32-35,80-84; PoseNet9D.py:39-41,49-50): max over the points of
    proj_layer(feat) = conv(LeakyReLU(BatchNorm(conv(feat)))), differentiable; feat (B, N, FEAT_C) rows"""
    B, N, C = feat.shape
    pl = enc.proj_layer
    x = linear(_pad4(feat.reshape(B * N, C)), _pad4(_w2(pl[0])))
    x = bn_act(x, pl[1], act=1, slope=0.2)
    y = linear(_pad4(x), _pad4(_w2(pl[3])))
    return colmax(y.view(B, N, -1))


def posenet_forward(net, points, obj_id, train_keys, sample_idx=None, inject=None, record=None, kmax=20, n_cls=6, cut=None,
                    enable_proj=False):
    """PoseNet9D.forward (PoseNet9D.py:33-91) with autograd; net is the drop-in module (training mode).
    cut: an EncoderCut to split the backward at the encoder's output `feat` (GraphedStep's two-segment form: everything after
    `feat` is differentiated first, the encoder afterwards, so that the late layers' gradients can travel meanwhile)."""
    B, N, _ = points.shape
    if B < 2:
        raise ValueError("Expected more than 1 value per channel when training, got input size [%d, 256]" % B)
    if sample_idx is None:
        sample_idx = engine.draw_sample_idx(N)
    points = points.contiguous().float()
    with torch.no_grad():                      # the clouds are data: no gradient flows to them
        xyz, mean = ops.center(points)         # bit-identical to the reference's centring (kNN indices depend on it)
        mean = mean.unsqueeze(1)
    if net.only_encoder:
        face = net.face_enc
        graphs = _GraphSource(points.device, inject, record, "face_enc.encoder.")
        feat, parts = encoder(face.encoder, xyz, obj_id.to(points.device), sample_idx, graphs, kmax, n_cls)
        dec0 = face.decoder.conv1d_block[0]
        xd = feat_consumers_factored(parts, [(_w2(dec0), dec0.bias)])[0] if parts is not None else None
        fo = feat[:, :, :FEAT_C] if parts is None else feat
        return dict(feat_global=proj_global(face.encoder, fo) if enable_proj else colmax(fo), recon=decoder(face.decoder, feat, None, xd))
    face = net.face_all
    graphs = _GraphSource(points.device, inject, record, "face_all.encoder.")
    feat, parts = encoder(face.encoder, xyz, obj_id.to(points.device), sample_idx, graphs, kmax, n_cls)
    if cut is not None:
        if parts is None:
            feat = cut.split(feat)[0]
        else:
            feat, fm01, fm23, fm4 = cut.split(feat, *parts[:3])
            parts = (fm01, fm23, fm4) + parts[3:]
    # the five layers over `feat` as one autograd node: factored over the up-sampling (_FeatConsumersFactored), or over the concat
    # buffer with d feat accumulated inside their dx GEMMs (_FeatConsumers)
    w1 = _w2
    dec0 = face.decoder.conv1d_block[0]
    layers = [(w1(face.ph_pred.conv_5[0]), None), (w1(dec0), dec0.bias), (w1(net.rot_green.conv1), net.rot_green.conv1.bias),
              (w1(net.rot_red.conv1), net.rot_red.conv1.bias), (w1(net.ts.conv1), net.ts.conv1.bias)]
    x5, xd, xg, xr, xt = feat_consumers(feat, layers) if parts is None else feat_consumers_factored(parts, layers)
    back, h1, h2 = ph_predictor(face.ph_pred, feat, x5)
    recon = decoder(face.decoder, feat, back, xd)
    green = point_head(net.rot_green, feat, xg)
    red = point_head(net.rot_red, feat, xr)
    ts = point_head(net.ts, feat, xt)
    out = dict()
    if train_keys:
        out["recon"] = recon + mean
    (out["p_green_R"], out["p_red_R"], out["f_green_R"], out["f_red_R"], out["Pred_T"], out["Pred_s"]) = _HeadPost.apply(
        green, red, ts, mean[:, 0].contiguous())
    if train_keys:
        out["h1"], out["h2"] = h1, h2
        out["feat"] = feat[:, :, :FEAT_C] if parts is None else feat
        out["feat_global"] = proj_global(face.encoder, out["feat"]) if enable_proj else colmax(out["feat"])
    return out


class EncoderCut(object):
    """Splits one backward pass at the encoder's output: the layers after `feat` see a detached copy that collects d loss / d feat;
    ``backward_encoder()`` then pushes that gradient through the encoder.  Two calls instead of one ``loss.backward()``, same
    gradients (the later layers read nothing of the encoder but what split() was given)."""

    def __init__(self):
        self.outs = self.leaves = None

    def split(self, *outs):
        """outs: the encoder's outputs that the later layers read (feat; with the factored layers also fine, [fm_2 | fm_3], fm_4)"""
        self.outs = outs
        self.leaves = tuple(t.detach().requires_grad_(True) for t in outs)
        return self.leaves

    def backward_encoder(self):
        pairs = [(t, l.grad) for t, l in zip(self.outs, self.leaves) if l.grad is not None]
        torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])

    def clear(self):
        self.outs = self.leaves = None


# parameters whose gradients are complete when the first backward segment (everything after the encoder) ends
LATE_PREFIXES = ("face_all.ph_pred.", "face_all.decoder.", "rot_green.", "rot_red.", "ts.")


class GraphedStep(object):
    """Forward(s) + loss + backward of one training step captured as a single hipGraph and replayed.

    ``step_fn(samples) -> scalar loss`` runs every forward of the step (the trainer's net1 with gradients and net2 under
    no_grad, trainer/RL_TDA.py:116-118) and the loss on STATIC input buffers it owns; ``samples[i] = (pool_1, pool_2)`` are the
    device-resident subsample indices of forward i (cloud size ``cloud_sizes[i]``), redrawn on the host before every replay in
    the reference's order (gcn3d.py:241-242: torch's global CPU generator, forward by forward) and copied in through a pinned
    ring.  The unfused training graph is ~700 kernel launches per step; replayed from a graph their launch overhead and the
    gaps between dependent kernels are gone.  Gradients land in the parameters' static ``.grad`` buffers (zeroed in place
    before each replay -- never set them to None afterwards, the graph writes to those very tensors); the gradient all-reduce,
    the clip and the optimizer step stay with the caller.  step_fn must be built from ops that can be captured (no host
    synchronisation).

    Losses / outputs of earlier EAGER steps over the same parameters must be dropped first: a live autograd graph keeps its
    AccumulateGrad nodes, which are bound to the stream they were created on, and a captured backward that handed its gradients
    to such a node would pull that (default) stream into the capture, which hipStreamEndCapture does not survive.  The
    constructor detects this during its warm-up backward -- outside any capture -- and raises RuntimeError instead of capturing
    (torch reports the foreign node as a warning; here it is an error).  The nodes the captured step itself needs are created
    inside the capture, on the capture stream, and freed with it (the returned loss is detached)."""

    _FOREIGN = "AccumulateGrad node's stream does not match"

    def __init__(self, params, step_fn, cloud_sizes, device, cut=None, between=None, after=None, own_pool=False, buckets=None):
        """cut: an EncoderCut that step_fn hands to net1's forward.  The step is then captured as TWO graphs sharing one memory
        pool -- (1) forwards + loss + backward of everything after the encoder, (2) the encoder's backward -- and replayed with
        ``between()`` called after the first has been enqueued (the data-parallel trainer starts the late layers' gradient
        exchange there: it overlaps the encoder's backward) and ``after()`` after the second."""
        import warnings
        dev = torch.device(device)
        self.params, self.sizes = list(params), [int(n) for n in cloud_sizes]
        self.counts = [(n // 4, n // 4 // 4) for n in self.sizes]
        total = sum(a + b for a, b in self.counts)
        self._pins = engine.PinnedRing(total)
        self._s12 = torch.zeros(total, dtype=torch.int32, device=dev)
        self.samples = [(torch.zeros(a, dtype=torch.int32, device=dev), torch.zeros(b, dtype=torch.int32, device=dev))
                        for a, b in self.counts]
        self.cut, self.between, self.after = cut, between, after
        self.buckets = buckets           # shard.GradBuckets whose flat buffers the gradients are views of (zeroed in two fills)

        def run():
            o = 0
            for (a, b), (s1, s2) in zip(self.counts, self.samples):
                s1.copy_(self._s12[o:o + a])
                s2.copy_(self._s12[o + a:o + a + b])
                o += a + b
            loss = step_fn(self.samples)
            loss.backward()
            return loss.detach()         # nothing keeps the step's autograd graph (and its AccumulateGrad nodes) alive

        def run2():
            cut.backward_encoder()

        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)      # the engine reports a foreign node once per process otherwise
        try:
            with torch.cuda.stream(warm), warnings.catch_warnings():
                warnings.filterwarnings("error", message=".*" + self._FOREIGN)
                for _ in range(2):
                    self._draw(None)
                    self._zero()
                    try:
                        run()
                        if cut is not None:
                            run2()
                            cut.clear()
                    except (UserWarning, RuntimeError) as e:
                        if self._FOREIGN not in str(e):
                            raise
                        raise RuntimeError(
                            "GraphedStep: an autograd graph of an earlier eager step over these parameters is still alive (a "
                            "loss or output tensor is being kept); its AccumulateGrad nodes belong to another stream and cannot "
                            "be captured.  Drop those tensors (del / .detach() / float()) before capturing the step.") from None
        finally:
            torch.set_warn_always(always)
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        self._zero()
        self.graph = engine.new_graph()
        with torch.cuda.graph(self.graph, stream=warm):     # the stream the warm-up ran on
            self.loss = run()
        # no memset node may be part of a captured step (engine.check_capture: the root cause of the round-2 replay fault)
        self.nodes = engine.check_capture(self.graph, "GraphedStep")
        self.graph2 = None
        if cut is not None:
            # the encoder's saved activations live in the first graph's pool: the second graph shares it and always replays after the first
            self.graph2 = engine.new_graph()
            with torch.cuda.graph(self.graph2, stream=warm, pool=None if own_pool else self.graph.pool()):
                run2()
            self.nodes2 = engine.check_capture(self.graph2, "GraphedStep (encoder backward segment)")
            cut.clear()

    def _zero(self):
        """gradients to zero before a replay: the flat buckets in two fills when the gradients are their views (overlap=True), else
        one multi-tensor launch (round 3 issued one fill per parameter: 164 launches in front of every replay)"""
        grads = [p.grad for p in self.params if p.grad is not None]
        if self.buckets is not None:
            self.buckets.zero_()
            lo_hi = [(f.data_ptr(), f.data_ptr() + 4 * f.numel()) for f in self.buckets.flat]
            grads = [g for g in grads if not any(lo <= g.data_ptr() < hi for lo, hi in lo_hi)]
        if grads:
            torch._foreach_zero_(grads)

    def _draw(self, sample_idx):
        if sample_idx is None:
            sample_idx = [engine.draw_sample_idx(n) for n in self.sizes]
        self._pins.upload(torch.cat([t.reshape(-1) for pair in sample_idx for t in pair]), self._s12)

    def __call__(self, sample_idx=None):
        self._draw(sample_idx)
        self._zero()
        self.graph.replay()
        if self.graph2 is not None:
            if self.between is not None:
                self.between()
            self.graph2.replay()
        if self.after is not None:
            self.after()
        return self.loss


class GraphedBackward(GraphedStep):
    """GraphedStep for a single network: ``loss_fn(net(points, obj_id)) -> scalar`` with the inputs in static buffers
    (bench.py's forward-plus-loss workloads and the tests; the trainer's two-network step is trainer/RL_TDA.graphed_step)."""

    def __init__(self, net, points, obj_id, loss_fn):
        self.net, self.N = net, points.shape[1]
        self.points, self.obj = points.clone(), obj_id.clone().float()
        super().__init__(net.parameters(), lambda samples: loss_fn(net(self.points, self.obj, sample_idx=samples[0])),
                         [self.N], points.device)

    def __call__(self, points=None, obj_id=None, sample_idx=None):
        if points is not None:
            self.points.copy_(points, non_blocking=True)
        if obj_id is not None:
            self.obj.copy_(obj_id.reshape(self.obj.shape).float(), non_blocking=True)
        return super().__call__(None if sample_idx is None else [sample_idx])
