"""Forward pipeline of PoseNet9D on the HIP kernels (host orchestration only): eval mode, and at the end of the
file the training-mode variant (batch-statistics BatchNorm, dropout).

This file turns a reference-format state dict into packed device weights (BatchNorm folded into
per-channel scale/shift, HS-layer projection and STE weights concatenated into one GEMM operand,
conv2 split into its feature / global halves, head weights padded to the concat buffer's row
stride) and walks the layers of ``Face_Enc`` / ``PH_Predictor`` / ``Face_Dec`` / the three heads
by enqueuing kernels through ``ops`` (the C ABI).  No torch compute op touches activations.

Data layout in HBM (fp32, row-major, rows = points):
  feat   (B*N, 1292)  concat buffer: fm_0 | fm_1 | up(fm_2) | up(fm_3) | up(fm_4) | one-hot(6) |
                      centred xyz (3) | 3 zero columns.  Every layer writes its slice in place
                      (FaceRecon.py:75 torch.cat never happens); the three heads and conv_5 /
                      decoder read it with K = 1292 against zero-padded weights, which also gives
                      Pose_Ts its 1289-channel input (PoseNet9D.py:63) without a second buffer.
  proj9  (B*n, 9*Cout) per HS layer: [centre | 7 support blocks | STE] from ONE GEMM.
"""
import math
import os

import torch

from . import ops

BN_EPS = 1e-5
FEAT_C = 1286
FEAT_LD = 1292          # 1286 + 3 (xyz) rounded up to a multiple of 4 floats (16-byte rows)
LEVEL_CH = ((128, 128), (128, 256), (256, 256), (256, 512))   # (Cin, Cout) of conv_1..conv_4


_FOLDS = None   # while a Packed is being built: list of (bn name, scale tensor, shift tensor) to refresh later


def _fold_values(sd, name):
    scale = sd[name + ".weight"] / torch.sqrt(sd[name + ".running_var"] + BN_EPS)
    return scale, sd[name + ".bias"] - sd[name + ".running_mean"] * scale


def _bn_fold(sd, name):
    """eval-mode BatchNorm as per-channel (scale, shift); registered so Packed.refold can refresh it in place"""
    scale, shift = _fold_values(sd, name)
    scale, shift = scale.contiguous(), shift.contiguous()
    if _FOLDS is not None:
        _FOLDS.append((name, scale, shift))
    return scale, shift


def _pad_cols(w, cols):
    out = torch.zeros(w.shape[0], cols, device=w.device, dtype=w.dtype)
    out[:, : w.shape[1]] = w
    return out


def _pad_rows(w, rows):
    out = torch.zeros((rows,) + tuple(w.shape[1:]), device=w.device, dtype=w.dtype)
    out[: w.shape[0]] = w
    return out


def _dev_sd(sd, device):
    return {k: v.detach().to(device=device, dtype=torch.float32) for k, v in sd.items() if v.dtype.is_floating_point}


def pack_encoder(sd, e, device):
    """Face_Enc weights under prefix `e` ('face_all.encoder.') -> list of per-layer dicts."""
    conv = []
    c0 = dict(C=128)
    c0["sdn"] = ops.normalize_dirs(sd[e + "conv_0.directions"])
    c0["ste"] = _pad_cols(sd[e + "conv_0.STE_layer.weight"][:, :, 0], 4)           # (128, 3 -> 4)
    w2 = sd[e + "conv_0.conv2.weight"][:, :, 0]
    c0["w1"], c0["w2"] = w2[:, :128].contiguous(), w2[:, 128:].contiguous()
    c0["w1_s"] = ops.split_w(c0["w1"])
    c0["w2t"] = c0["w2"].t().contiguous()
    # conv2's feature half and the STE convolution as ONE operand over [g | x y z 0] (K = 132): the graph convolution writes the
    # point behind its 128 outputs (tgp_gconv_surface_fwd xyz_pad), so STE(xyz) costs four K columns instead of a fill, a copy
    # and a GEMM launch of its own (30 us of the serial forward)
    c0["w1x"] = torch.cat([c0["w1"], c0["ste"]], dim=1).contiguous()                   # (128, 132)
    c0["w1x_s"] = ops.split_w(c0["w1x"])
    if ops.planes_on():
        c0["w1x_p"] = ops.planes_w(c0["w1x"])
    conv.append(c0)
    for i, (cin, cout) in enumerate(LEVEL_CH, start=1):
        p = e + "conv_%d." % i
        c = dict(C=cout, Cin=cin)
        c["sdn"] = ops.normalize_dirs(sd[p + "directions"])
        # one GEMM operand: rows [weights^T (8*Cout) ; STE (Cout)], bias [bias ; 0]
        c["wcat"] = torch.cat([sd[p + "weights"].t(), sd[p + "STE_layer.weight"][:, :, 0]], dim=0).contiguous()
        c["wcat_s"] = ops.split_w(c["wcat"])
        if ops.planes_on():
            c["wcat_p"] = ops.planes_w(c["wcat"])
            if c["wcat_p"].tgp_unscale is None:                 # (a weight inside fp16's range: no pack-time rescale)
                c["wcat_u"] = ops.proj_pack(c["wcat"])
        c["bcat"] = torch.cat([sd[p + "bias"], torch.zeros(cout, device=device)]).contiguous()
        w2 = sd[p + "conv2.weight"][:, :, 0]
        c["w1"], c["w2"] = w2[:, :cout].contiguous(), w2[:, cout:].contiguous()
        c["w1_s"] = ops.split_w(c["w1"])
        if ops.planes_on():
            c["w1_p"] = ops.planes_w(c["w1"])
        c["w2t"] = c["w2"].t().contiguous()
        if i <= 3:
            c["scale"], c["shift"] = _bn_fold(sd, e + "bn%d" % i)
        conv.append(c)
    return conv


def pack_ph(sd, ph):
    lin = lambda n: (sd[ph + n + ".weight"].contiguous(), sd[ph + n + ".bias"].contiguous())
    l2, l3, l4, l5 = lin("linear2"), lin("linear3"), lin("linear4"), lin("linear5")
    return dict(w5=_pad_cols(sd[ph + "conv_5.0.weight"][:, :, 0], FEAT_LD), bn5c=_bn_fold(sd, ph + "conv_5.1"),
                l1=sd[ph + "linear1.weight"].contiguous(), bn5=_bn_fold(sd, ph + "bn5"),
                # linear2 | linear3 share their input: one (5000, 1024) operand; linear4(pi1) + linear5(pi2) is one
                # (1286, 5000) operand applied to [pi1 | pi2] with bias b4 + b5
                l23=(torch.cat([l2[0], l3[0]], 0).contiguous(), torch.cat([l2[1], l3[1]]).contiguous()),
                l45=(torch.cat([l4[0], l5[0]], 1).contiguous(), (l4[1] + l5[1]).contiguous()), n_code=l2[0].shape[0])


def pack_decoder(sd, d):
    dec = []
    for conv, bn in (("conv1d_block.0", "conv1d_block.1"), ("conv1d_block.3", "conv1d_block.4"),
                     ("conv1d_block.6", "conv1d_block.7"), ("recon_head.0", "recon_head.1")):
        w = sd[d + conv + ".weight"][:, :, 0]
        if w.shape[1] == FEAT_C:
            w = _pad_cols(w, FEAT_LD)
        w = w.contiguous()
        dec.append((w, sd[d + conv + ".bias"].contiguous()) + _bn_fold(sd, d + bn) + (ops.split_w(w),)
                   + (ops.planes_w(w) if ops.planes_on() and w.shape[1] != FEAT_LD else None,))
    return dec, (sd[d + "recon_head.3.weight"][:, :, 0].contiguous(), sd[d + "recon_head.3.bias"].contiguous())


def pack_head(sd, h):
    """Rot_green / Rot_red / Pose_Ts under prefix `h` ('rot_green.'); conv1 padded to the concat stride."""
    cv = lambda n: (sd[h + n + ".weight"][:, :, 0].contiguous(), sd[h + n + ".bias"].contiguous())
    w1, b1 = cv("conv1")
    return dict(k_alg=w1.shape[1], c1=(_pad_cols(w1, FEAT_LD), b1) + _bn_fold(sd, h + "bn1"),
                c2=cv("conv2") + _bn_fold(sd, h + "bn2"),
                c3=cv("conv3") + _bn_fold(sd, h + "bn3"), c4=cv("conv4"))


HEAD_ORDER = ("rot_green", "rot_red", "ts")


def pack_wide(ph, heads, bn_names=None):
    """conv_5 (PH_Predictor) and the three heads' conv1 all read `feat`: one (4096, 1292) operand.
    Columns [0,1024) = conv_5 (no bias, LeakyReLU 0.2, only its max over points is needed);
    columns [1024,4096) = rot_green | rot_red | ts conv1 (+bias, ReLU).  Also stacks the heads' conv2."""
    dev = ph["w5"].device
    z = torch.zeros(1024, device=dev)
    cat = lambda i: torch.cat([hd["c1"][i] for hd in heads]).contiguous()
    c2 = lambda i: torch.stack([hd["c2"][i] for hd in heads]).contiguous()
    w = dict(
        W=torch.cat([ph["w5"]] + [hd["c1"][0] for hd in heads], dim=0).contiguous(),
        bias=torch.cat([z, cat(1)]).contiguous(),
        scale=torch.cat([ph["bn5c"][0], cat(2)]).contiguous(),
        shift=torch.cat([ph["bn5c"][1], cat(3)]).contiguous(),
        slope=torch.cat([z + 0.2, torch.zeros(3072, device=dev)]).contiguous(),
        k_alg=(FEAT_C + sum(hd["k_alg"] for hd in heads)) / 4.0,
        W2=c2(0), b2=c2(1), scale2=c2(2), shift2=c2(3))
    w["Ws"] = ops.split_w(w["W"])
    w["W2s"] = ops.split_w(w["W2"])
    # conv3 / conv4 of the three heads as batched skinny GEMMs (conv4 rows padded to 8: outputs 4, 4, 6)
    c3 = lambda i: torch.stack([hd["c3"][i] for hd in heads]).contiguous()
    w["W3"], w["b3"], w["scale3"], w["shift3"] = c3(0), c3(1), c3(2), c3(3)
    w["W4"] = torch.stack([_pad_rows(hd["c4"][0], 8) for hd in heads]).contiguous()
    w["b4"] = torch.stack([_pad_rows(hd["c4"][1].unsqueeze(1), 8)[:, 0] for hd in heads]).contiguous()
    w["n_out"] = [hd["c4"][0].shape[0] for hd in heads]
    w["W3t"] = w["W3"].transpose(1, 2).contiguous() if tuple(w["W3"].shape) == (3, 256, 256) else None     # ops.pose_tail
    if _FOLDS is not None and bn_names is not None:      # the concatenated copies must follow the running statistics too
        _FOLDS.append((bn_names["conv5"], w["scale"][0:1024], w["shift"][0:1024]))
        for i, h in enumerate(bn_names["heads"]):
            _FOLDS.append((h + "bn1", w["scale"][1024 * (i + 1):1024 * (i + 2)], w["shift"][1024 * (i + 1):1024 * (i + 2)]))
            _FOLDS.append((h + "bn2", w["scale2"][i], w["shift2"][i]))
            _FOLDS.append((h + "bn3", w["scale3"][i], w["shift3"][i]))
    return w


class Packed(object):
    """Device-resident, kernel-ready weights of one PoseNet9D.  The heavy operands (concatenated / padded / bf16-split
    weights) depend on the parameters only; the eval-mode BatchNorm folds also depend on the running statistics and
    are refreshed in place by refold() when those change (every training step moves them)."""

    def __init__(self, sd, device, face="face_all.", with_heads=True):
        global _FOLDS
        sd = _dev_sd(sd, device)
        self.device, self.face = device, face
        self.folds = _FOLDS = []
        try:
            self.conv = pack_encoder(sd, face + "encoder.", device)
            self.ph = pack_ph(sd, face + "ph_pred.")
            self.dec, self.dec_out = pack_decoder(sd, face + "decoder.")
            # (round 4) The topology code reaches the decoder only as a per-object bias of its first conv:
            # conv(feat + back) = conv(feat) + W0 back with back = L45 [pi1 | pi2] + b45 (FaceRecon.py:158-165,112), so
            # W0 back = (W0 L45) pi + W0 b45: one (512, 5000) vector layer instead of the (1286, 5000) one and the (512, 1286)
            # one behind it.  The product of the two weights is taken in fp64 and rounded once.
            w0d, (w45, b45) = self.dec[0][0][:, :FEAT_C].double(), self.ph["l45"]
            self.rb_w = ((w0d @ w45.double()).float().contiguous(), (w0d @ b45.double()).float().contiguous())
            self.heads = {h: pack_head(sd, h + ".") for h in HEAD_ORDER} if with_heads else {}
            if with_heads:
                self.wide = pack_wide(self.ph, [self.heads[h] for h in HEAD_ORDER],
                                      dict(conv5=face + "ph_pred.conv_5.1", heads=[h + "." for h in HEAD_ORDER]))
                self.fact = pack_factored(self.wide, self.dec[0])
            # the fused decoder kernel's staging image of the three inner weights (fp16 operands without a pack-time rescale)
            # the staging images of ops.hs_chain (conv_0 / conv_2's last GEMM + the next layer's projection), when the weights lie
            # inside fp16's range (no pack-time rescale) and the layers have the two shapes the kernel serves
            self.chains = [None, None]
            if ops.planes_on():
                for i, (a, b, wk) in enumerate(((0, 1, "w1x"), (2, 3, "w1"))):
                    ca, cb = self.conv[a], self.conv[b]
                    pa, pb = ca.get(wk + "_p"), cb.get("wcat_p")
                    if pa is not None and pb is not None and pa.tgp_unscale is None and pb.tgp_unscale is None and "w2t" in ca:
                        self.chains[i] = ops.hs_chain_pack(ca[wk], cb["wcat"])
            self.dec_units = None
            if (ops.planes_on() and [tuple(d[0].shape) for d in self.dec[1:]] == [(512, 512), (256, 512), (128, 256)]
                    and tuple(self.dec_out[0].shape) == (3, 128) and all(d[5] is not None and d[5].tgp_unscale is None for d in self.dec[1:])):
                self.dec_units = ops.dec_pack(self.dec[1][0], self.dec[2][0], self.dec[3][0])
                # ... and, with the first conv on its own heads-style kernel (ops.dec_l1), whose result arrives in accumulator order
                self.dec_units_p = ops.dec_pack(self.dec[1][0], self.dec[2][0], self.dec[3][0], h1_permuted=True)
        finally:
            _FOLDS = None

    def refold(self, sd):
        sd = _dev_sd(sd, self.device)
        for name, scale, shift in self.folds:
            sc, sh = _fold_values(sd, name)
            scale.copy_(sc)
            shift.copy_(sh)
        f = getattr(self, "fact", None)
        if f is not None and f.get("w2p") is not None:      # the fused heads kernel reads conv1's fold from inside its packed operand
            w = self.wide
            ops.heads_repack_vectors(f["w2p"], w["bias"][1024:], w["scale"][1024:], w["shift"][1024:])


PROJ_LD = 1288          # FEAT_C padded to whole float4: row stride of the projection head's inner activation


def proj_operands(sd, face, device):
    """Face_Enc.proj_layer (FaceRecon.py:32-35; applied to feat_global when enable_proj, :80-84) as GEMM operands over the concat
    buffer: Conv1d 1286 -> 1286 (no bias) | BatchNorm1d | LeakyReLU(0.2) | Conv1d 1286 -> 1286 (no bias).  Rows / columns are
    zero-padded to the buffers' strides (1292 in, 1288 between and out).  Built on first use: the reference's callers never pass
    enable_proj (its 3.31 M parameters are in every checkpoint all the same)."""
    pl = face + "encoder.proj_layer."
    sd = _dev_sd({k: v for k, v in sd.items() if k.startswith(pl)}, device)
    w1 = _pad_rows(_pad_cols(sd[pl + "0.weight"][:, :, 0], FEAT_LD), PROJ_LD).contiguous()
    w2 = _pad_rows(_pad_cols(sd[pl + "3.weight"][:, :, 0], PROJ_LD), PROJ_LD).contiguous()
    return dict(name=pl + "1", w1=w1, w2=w2, w1s=ops.split_w(w1), w2s=ops.split_w(w2))


def proj_feat_global(po, feat, sd=None, bn=None):
    """feat_global with enable_proj=True (PoseNet9D.py:39-41,49-50): max over each object's points of proj_layer(feat).  feat: the
    concat buffer (B, N, FEAT_LD).  bn: the training-mode BatchNorm applier (batch statistics, moves the buffers); else the eval
    fold of the running statistics in sd.  The second conv's activation is never stored: its epilogue keeps the maxima as keys."""
    B, N, ld = feat.shape
    if ld != FEAT_LD or feat.stride(1) != FEAT_LD:
        raise ValueError("proj_feat_global: the concat buffer (row stride %d) expected" % FEAT_LD)
    M = B * N
    dev = feat.device
    f2 = feat.view(M, FEAT_LD)
    if bn is None:
        scale, shift = _fold_values(sd, po["name"])
        one, zero = torch.ones(PROJ_LD - FEAT_C, device=dev), torch.zeros(PROJ_LD - FEAT_C, device=dev)
        h = ops.linear_rows(f2, po["w1"], scale=torch.cat([scale.to(dev), one]), shift=torch.cat([shift.to(dev), zero]), act=1, slope=0.2,
                            w_split=po["w1s"])
    else:
        h = ops.linear_rows(f2, po["w1"], w_split=po["w1s"])
        bn(po["name"], h[:, :FEAT_C], act=1, slope=0.2)
    keys = torch.zeros(B, PROJ_LD, device=dev, dtype=torch.int32)
    ops.linear_rows(h, po["w2"], colmax_keys=keys, rows_per_obj=N, want_out=False, w_split=po["w2s"])
    return ops.colmax_decode(keys)[:, :FEAT_C].contiguous()


FINE_K = 268            # fm_0 | fm_1 | one-hot | xyz | 3 zero columns: the part of `feat` that differs from point to point
FINE_LD = 272


def _fine_cols(w):
    """(rows, FEAT_LD) weight over the concat buffer -> its columns over the fine part, (rows, FINE_LD) zero padded"""
    return _pad_cols(torch.cat([w[:, :256], w[:, 1280:FEAT_LD]], dim=1), FINE_LD).contiguous()


def pack_factored(wide, dec0):
    """Operands of the factored form of the two layers that read `feat` = [fm_0 | fm_1 | up(fm_2) | up(fm_3) | up(fm_4) | tail]
    (the fused conv_5 + head conv1 GEMM and the decoder's first conv).  up() is nearest-neighbour upsampling
    (FaceRecon.py:70-75), so W x feat = W_fine x fine + (W_1 x [fm_2 | fm_3])[near_1] + (W_2 x fm_4)[near_2]: the two coarse
    products are taken once per COARSE point (N/4 and N/16 of them) and fetched per point by the fine GEMM's epilogue.
    The coarse operands of both layers are stacked (4096 + 512 rows) so each level is one launch."""
    W, w0 = wide["W"], dec0[0]
    f = dict(Wa=_fine_cols(W), dec_a=_fine_cols(w0),
             Wb=torch.cat([W[:, 256:768], w0[:, 256:768]], dim=0).contiguous(),
             Wc=torch.cat([W[:, 768:1280], w0[:, 768:1280]], dim=0).contiguous())
    for k in ("Wa", "dec_a", "Wb", "Wc"):
        f[k + "_s"] = ops.split_w(f[k])
    if ops.planes_on():
        for k in ("dec_a", "Wb", "Wc"):
            f[k + "_p"] = ops.planes_w(f[k])
        for k in ("Wb", "Wc"):                                  # the coarse products on the projection kernel (ops.proj_planes)
            f[k + "_u"] = ops.proj_pack(f[k]) if f[k + "_p"].tgp_unscale is None else None
    # the fused heads kernel (conv1 -> conv2 -> max in one launch) takes fp16 operands without a pack-time rescale
    f["w2p"] = f["Wa_hp"] = f["Wa_cp"] = None
    if (ops.GEMM_MODE == "split16" and getattr(f["Wa_s"], "tgp_unscale", None) is None
            and float(wide["W2"].abs().max()) < ops.FP16_SAFE):
        f["w2p"] = ops.heads_pack_w2(wide["W2"], wide["bias"][1024:], wide["scale"][1024:], wide["shift"][1024:])
        f["Wa_hp"] = ops.heads_planes_w(f["Wa"][1024:])
        f["Wa_cp"] = ops.heads_planes_w(f["Wa"][:1024])         # conv_5's rows, for tgp_conv_max_fused
    f["dec_a_hp"] = ops.heads_planes_w(f["dec_a"]) if ops.planes_on() and f["dec_a"].shape[0] == 512 else None
    return f


def _i32(idx, device):
    """injected graph (any int dtype, any device, (B,n,k) or (B,n,1)) -> contiguous int32 on device"""
    return idx.to(device=device, dtype=torch.int32).contiguous()


_PIN = {}
_SIDE = {}
BRANCH_STREAMS = True   # run the PH-tail -> decoder chain beside the head chain on a second HIP stream
FACTORED = True         # eval forward: the layers over the concat buffer run factored over the upsampling (pack_factored)
HEADS_FUSED = True      # ... and the heads' conv1 -> conv2 -> max as one kernel (csrc/heads_fused.hip) instead of two GEMM launches
# Two more side branches, built and measured in round 3 (scripts/branch_ab.py; profiles/r03_branch_ab.txt) and
# left OFF: the level-1 coarse product beside conv_4, and the rows behind the fused heads kernel's last full round (771 workgroups
# = 3.01 rounds of 256) on the tile kernels beside it.  One batch in flight: +1 .. 3 % (a chip-filling GEMM beside a chain of small
# launches does not shorten the chain -- its workgroups hold every CU, and the small launches wait for them).  Two batches in
# flight (the bench default): 17.7 k -> 15.3 k objects/s.  hipGraph runs a graph's extra branches on ONE pool of internal streams
# per device shared by every graph in flight (DEBUG_HIP_FORCE_GRAPH_QUEUES / GPU_MAX_HW_QUEUES = 8 / 16 changed nothing), so the
# branches of two replays queue behind each other.
# (Round 4, the same for a third extra branch -- the up-sampling look-ups, the row sort and the sorted fine buffer beside conv_4, 70 us of
# launches that need only coordinates and fm_0 / fm_1: one batch in flight 15.70 -> 15.74 k alone, 15.83 k with COARSE_SIDE alone, 15.36 k
# with both.  Not kept.)
COARSE_SIDE = os.environ.get("TGP_COARSE_SIDE", "0") != "0"
# Eval forward without the layers whose results the six-key eval dict does not return (PH predictor, decoder: 30 % of the
# reference's FLOPs).  A deployment switch; the bench's headline and every parity test run the full forward.
EVAL_OUTPUTS_ONLY = os.environ.get("TGP_EVAL_OUTPUTS_ONLY", "0") != "0"
HEADS_TAIL = os.environ.get("TGP_HEADS_TAIL", "0") != "0"
DEC_PLANES_ONLY = os.environ.get("TGP_DEC_PLANES_ONLY", "1") != "0"     # the decoder's inner activations as fp16 planes only
DEC_L1 = os.environ.get("TGP_DEC_L1", "1") != "0"           # the decoder's first conv on the fused heads kernel's conv1 half (ops.dec_l1)
DEC_FUSED = os.environ.get("TGP_DEC_FUSED", "1") != "0"     # ... and everything behind its first conv as one launch (csrc/dec_fused.hip)
PROJ_KERNEL = os.environ.get("TGP_PROJ_KERNEL", "1") != "0"      # the HS layers' projection GEMMs on tgp_proj_planes (csrc/hs_chain.hip)
# conv_0's / conv_2's last GEMM + the next layer's projection as one launch (csrc/hs_chain.hip, bit-identical results).  Built and measured
# in round 5, OFF by default: alone the pairs take 71 / 70 us against 83-94 / 73-86 us as two launches, but the kernel holds its CUs
# exclusively (one wave per SIMD, 150 KB of LDS) while it waits for its 186 MB of stores, and on the four-in-flight line the same-box A/B
# shows no gain (21.15 k against 21.44 k objects/s, three alternating runs); it also adds two predicated launches per forward.
HS_CHAIN = os.environ.get("TGP_HS_CHAIN", "0") != "0"
REPAIR_OBJS = 16        # objects per chunk of the fused heads kernel's fp16-range repair (wide_gemm_factored)


SIDE_TAG = 0            # GraphedForward gives each half batch its own side stream


def _side_stream(device, tag=None):
    key = (torch.device(device).index, SIDE_TAG if tag is None else tag)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]



def _upload_i32(t, device):
    """small host index vector -> device int32 through a reused pinned staging buffer (asynchronous copy; a
    pageable-memory copy would block the host until the stream drains and leave the GPU idle between forwards)"""
    if t.is_cuda:
        return t.to(device=device, dtype=torch.int32).contiguous()
    n = t.numel()
    ring = _PIN.setdefault(n, dict(slots=[[torch.empty(n, dtype=torch.int32).pin_memory(), None] for _ in range(4)], i=0))
    ring["i"] = (ring["i"] + 1) % len(ring["slots"])
    slot = ring["slots"][ring["i"]]
    if slot[1] is not None:
        slot[1].synchronize()          # the copy that last used this staging buffer has completed
    slot[0].copy_(t)
    out = slot[0].to(device, non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(device))
    return out


class Graphs(object):
    """Neighbour lists of one forward.  Computed lists are shared between the call sites that the
    reference recomputes identically (gcn3d.py:85/213/235 all run the same xyz kNN); a test may
    inject the reference's own lists per call site (names as in tests/golden)."""

    def __init__(self, device, inject=None, record=None, prefix="face_all.encoder."):
        self.device, self.inject, self.record, self.prefix = device, inject or {}, record, prefix

    def get(self, name, compute):
        full = self.prefix + name
        if full in self.inject:
            idx = _i32(self.inject[full], self.device)
        else:
            idx = compute()
        if self.record is not None:
            self.record[full] = idx
        return idx


def _chain_ok(chain, gp, out_p, M, act):
    """may this layer's last GEMM run together with the next layer's projection (ops.hs_chain)?  Only in a forward without side
    branches (with them the projection runs beside the feature-space kNN, which needs this layer's output first), on planes, with the
    weights inside fp16's range, and for row counts at which BOTH launches it replaces run on the tile kernels -- it computes their
    bits, not the exact-fp32 small kernel's."""
    if chain is None or not HS_CHAIN or BRANCH_STREAMS or gp is None or out_p is None or chain.get("units") is None or act is None:
        return False
    _, N1, N2 = chain["units"].tgp_shape
    return ops._routes_to_big_tile(M, N1, 1, True) and ops._routes_to_big_tile(M, N2, 1, True)


def _layer_tail_chained(chain, x, w, last, col0):
    """the layer's last GEMM (`last`: its ops.linear_rows arguments) + the next layer's projection in one launch; the two launches
    follow predicated on the kernel's fp16-range flag (they normally return at once).  Leaves the projection in chain["proj9"]."""
    nxt, flag = chain["next"], chain["flag"]
    out = last["out"]
    B, n = out.shape[0], out.shape[1]
    proj9 = torch.empty(B, n, nxt["wcat"].shape[0], device=out.device, dtype=torch.float32)
    ops.hs_chain(last["a_planes"], chain["units"], out, nxt["bcat"], flag, rowbias=last["rowbias"], rows_per_obj=last["rows_per_obj"],
                 res1=last["res1"], res2=last.get("res2"), scale1=last["scale"], shift1=last["shift"], relu=True, c1_planes=last["c_planes"],
                 c1_col0=col0, c2=proj9.view(B * n, -1))
    ops.linear_rows(x, w, pred=flag, **last)
    ops.linear_rows(out, nxt["wcat"], bias=nxt["bcat"], out=proj9, w_split=nxt.get("wcat_s"), a_planes=_cols_view(last["c_planes"], col0, out.shape[-1]),
                    w_planes=nxt.get("wcat_p"), pred=flag)
    chain["proj9"] = proj9
    return out


def _cols_view(pl, col0, K):
    """the planes of columns [col0, col0 + K) of a wider planes buffer as the consumer sees them (same buffer and magnitude words; the
    projection reads the first K columns when col0 == 0)"""
    if col0 != 0:
        raise ValueError("chained layers write their planes at column 0 of the next layer's operand")
    return pl


def surface_layer(c, xyz, idx_rf, idx_orl, out, scale=None, shift=None, act=None, out_p=None, amax_ws=None, tickets=None, chain=None):
    """HSlayer_surface.forward (gcn3d.py:78-89) + the caller's activation, written to `out` (B,n,C) view (and, out_p, as the fp16
    planes the next layer's projection GEMM stages by LDS-DMA)."""
    B, n, _ = xyz.shape
    C = c["C"]
    gx = torch.empty(B, n, C + 4, device=xyz.device, dtype=torch.float32)          # [g | x y z 0]
    g = ops.gconv_surface(xyz, idx_rf, c["sdn"], 7, C, out=gx[:, :, :C], xyz_pad=True)
    gp = None
    if tickets is None and ops.ORL_FUSED and "w2t" in c:      # (outside the arena: the concat-buffer path) the same kernel, so the same bits
        tickets = torch.zeros(B, device=xyz.device, dtype=torch.int32)
    if out_p is not None and "w1x_p" in c and "w2t" in c:
        # the ORL pooling stages g in LDS anyway: it leaves [g | x y z 0] as fp16 planes, the last GEMM's operand (csrc/gconv.hip)
        rb, gp = ops.orl_rowbias(g, idx_orl, c["w2t"], planes=ops.Planes(B * n, C + 4, xyz.device, amax_buf=amax_ws), xyz_tile=xyz,
                                 tickets=tickets)
    else:
        rb = ops.orl_rowbias(g, idx_orl, c["w2t"], tickets=tickets) if "w2t" in c else ops.linear_rows(ops.orl_global(g, idx_orl), c["w2"])
    # conv2(cat[g, global]) + g + STE(xyz) (gcn3d.py:87-89,108-112): W1 g + STE xyz is one product over the padded row
    last = dict(out=out, rowbias=rb, rows_per_obj=n, res1=g, scale=scale, shift=shift, act=0 if act is None else 1, slope=0.0,
                w_split=c.get("w1x_s"), k_alg=C + 3, c_planes=out_p, a_planes=gp, w_planes=c.get("w1x_p") if gp is not None else None)
    if _chain_ok(chain, gp, out_p, B * n, act):
        return _layer_tail_chained(chain, gx, c["w1x"], last, 0)
    ops.linear_rows(gx, c["w1x"], **last)
    return out


def _beside(device, fn, tag="knn"):
    """run fn() on a side stream forked from the current one; returns (result, join event).  Inside a captured graph this
    is a parallel branch; in eager mode the allocator is told that the results will be consumed on the current stream."""
    cur = torch.cuda.current_stream(device)
    side = _side_stream(device, (SIDE_TAG, tag))
    fork = torch.cuda.Event()
    fork.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(fork)
        res = fn()
        join = torch.cuda.Event()
        join.record(side)
    if not torch.cuda.is_current_stream_capturing():
        for t in (res if isinstance(res, (tuple, list)) else (res,)):
            if torch.is_tensor(t):
                t.record_stream(cur)
    return res, join


def hs_layer(c, xyz, fmap, idx_rf, idx_orl, out, scale=None, shift=None, act=None, fmap_p=None, out_p=None, out_col0=0, amax_ws=None,
             tickets=None, chain=None, proj9=None):
    """HS_layer.forward (gcn3d.py:142-155) + the caller's BatchNorm(eval)/ReLU, written to `out`.
    idx_rf / idx_orl may be callables: they are then evaluated on a side stream (the feature-space kNN -- distance GEMM +
    selection -- and the level's xyz kNN depend only on the layer's inputs) while this stream runs the projection GEMM."""
    B, n, _ = xyz.shape
    C = c["C"]
    join = None
    dirs = None
    if callable(idx_rf) or callable(idx_orl):
        def lists():
            # (idx_rf() may return (idx, dirs): ops.knn_feat(xyz=...) -- flat, so that _beside sees every tensor)
            rf = idx_rf() if callable(idx_rf) else idx_rf
            rf, dr = rf if isinstance(rf, tuple) else (rf, None)
            return rf, dr, (idx_orl() if callable(idx_orl) else idx_orl)
        if BRANCH_STREAMS:
            (idx_rf, dirs, idx_orl), join = _beside(xyz.device, lists)
        else:
            idx_rf, dirs, idx_orl = lists()
    # (B,n,9C): centre|support|STE.  fmap_p: the input's fp16 planes, written by its producer -- the GEMM then runs on the
    # pre-split kernel (csrc/gemm_pp.hip), bit-identical results
    if proj9 is None:         # (else: computed with the previous layer's last GEMM, ops.hs_chain)
        if (PROJ_KERNEL and fmap_p is not None and c.get("wcat_u") is not None and c["wcat_u"].tgp_shape[0] == fmap.shape[-1]
                and ops._routes_to_big_tile(B * n, c["wcat"].shape[0], 1, True)):
            # (round 5) the projection on its own kernel: the operand's fragments stay in registers for all 9 C columns (csrc/hs_chain.hip)
            proj9 = ops.proj_planes(fmap_p, c["wcat_u"], c["bcat"], fmap, c["wcat"])
        else:
            proj9 = ops.linear_rows(fmap, c["wcat"], bias=c["bcat"], w_split=c.get("wcat_s"), a_planes=fmap_p, w_planes=c.get("wcat_p"))
    if join is not None:
        torch.cuda.current_stream(xyz.device).wait_event(join)
    g = ops.gconv_hs(xyz, idx_rf, proj9, c["sdn"], 7, C, dirs=dirs)
    gp = None
    if tickets is None and ops.ORL_FUSED and "w2t" in c:
        tickets = torch.zeros(B, device=xyz.device, dtype=torch.int32)
    if amax_ws is not None and "w1_p" in c and "w2t" in c:
        rb, gp = ops.orl_rowbias(g, idx_orl, c["w2t"], planes=ops.Planes(B * n, C, xyz.device, amax_buf=amax_ws), tickets=tickets)
    else:
        rb = ops.orl_rowbias(g, idx_orl, c["w2t"], tickets=tickets) if "w2t" in c else ops.linear_rows(ops.orl_global(g, idx_orl), c["w2"])
    last = dict(out=out, rowbias=rb, rows_per_obj=n, res1=g, res2=proj9[:, :, 8 * C:], scale=scale, shift=shift,
                act=0 if act is None else 1, slope=0.0, w_split=c.get("w1_s"), c_planes=out_p, cp_col0=out_col0, a_planes=gp,
                w_planes=c.get("w1_p") if gp is not None else None)
    if _chain_ok(chain, gp, out_p, B * n, act):
        return _layer_tail_chained(chain, g, c["w1"], last, out_col0)
    ops.linear_rows(g, c["w1"], **last)
    return out


def encoder_forward(pk, points_c, obj_id, sample_idx, graphs, kmax=20, n_cls=6, factored=False, arena=None):
    """Face_Enc.forward (FaceRecon.py:39-86) -> feat buffer (B,N,FEAT_LD) and intermediates.
    factored: the concat buffer is not built; returns the fine buffer (B,N,FINE_LD) = fm_0 | fm_1 | one-hot | xyz | 0 and,
    in the intermediates, fm23 (B,N1,512), fm_4 and the absolute rows of each point's nearest coarse points."""
    dev = points_c.device
    B, N, _ = points_c.shape
    xyz = points_c
    feat = torch.empty(B, N, FINE_LD if factored else FEAT_LD, device=dev, dtype=torch.float32)
    N1, N2 = sample_idx[0].numel(), sample_idx[1].numel()
    s12 = _upload_i32(torch.cat([sample_idx[0].reshape(-1), sample_idx[1].reshape(-1)]), dev)
    s1, s2 = s12[:N1], s12[N1:]

    knn0 = {}

    def xyz_graph(level, pts, k):
        if level not in knn0:
            knn0[level] = ops.knn_xyz(pts, k)
        return knn0[level]

    cv = pk.conv
    # (round 4) activations that feed GEMMs also travel as blocked fp16 planes (ops.Planes), written by their producers: the
    # consuming GEMMs then stage both operands by LDS-DMA and never convert in their K loop (csrc/gemm_pp.hip).  Their per-block
    # magnitudes (the consumers' fp16 range guard) live in the forward's arena, zeroed by its one fill.
    pl = arena.planes(B, N, N1, N2) if (arena is not None and factored and ops.planes_on()) else {}
    fm0 = feat[:, :, 0:128]
    # (round 5) conv_0's and conv_2's last GEMM run together with the next layer's projection when the forward has no side branches
    chains = getattr(pk, "chains", [None, None]) if (pl and arena is not None) else [None, None]
    ch01 = dict(units=chains[0], next=cv[1], flag=arena.chain_flags[0]) if chains[0] is not None else None
    ch23 = dict(units=chains[1], next=cv[3], flag=arena.chain_flags[1]) if chains[1] is not None else None
    surface_layer(cv[0], xyz, graphs.get("conv_0.rf", lambda: xyz_graph(0, xyz, kmax)),
                  graphs.get("conv_0.orl_xyz", lambda: xyz_graph(0, xyz, kmax)), fm0, act="relu", out_p=pl.get("fm0"), amax_ws=pl.get("amax_g0"), tickets=pl.get("tick0"),
                  chain=ch01)
    fm1 = feat[:, :, 128:256]
    hs_layer(cv[1], xyz, fm0, lambda: graphs.get("conv_1.rf", lambda: ops.knn_feat(fm0, kmax)),
             graphs.get("conv_1.orl_xyz", lambda: xyz_graph(0, xyz, kmax)), fm1, cv[1]["scale"], cv[1]["shift"], "relu",
             fmap_p=pl.get("fm0"), amax_ws=pl.get("amax_g1"), tickets=pl.get("tick1"), proj9=(ch01 or {}).get("proj9"))
    v1, fp1 = ops.pool(xyz, fm1, graphs.get("pool_1.xyz", lambda: xyz_graph(0, xyz, kmax)), s1, kpool=4, planes=pl.get("fp1"))

    k1 = min(kmax, N1 // 8)
    fm23 = torch.empty(B, N1, 512, device=dev, dtype=torch.float32)      # fm_2 | fm_3 side by side: one GEMM operand when factored
    fm2, fm3 = fm23[:, :, :256], fm23[:, :, 256:]
    def feat_graph(name, fmap, k, pts):
        """the layer's feature-space list; at the pooled levels (LDS-staged graph convolution) with the unit directions to the
        selected neighbours, written by the selecting kernel -- unless the lists are injected / recorded (tests)"""
        if graphs.inject or graphs.record is not None or not KNN_DIRS:
            return graphs.get(name, lambda: ops.knn_feat(fmap, k))
        return ops.knn_feat(fmap, k, xyz=pts)

    hs_layer(cv[2], v1, fp1, lambda: feat_graph("conv_2.rf", fp1, k1, v1),
             lambda: graphs.get("conv_2.orl_xyz", lambda: xyz_graph(1, v1, k1)), fm2, cv[2]["scale"], cv[2]["shift"], "relu",
             fmap_p=pl.get("fp1"), out_p=pl.get("fm23"), out_col0=0, amax_ws=pl.get("amax_g2"), tickets=pl.get("tick2"), chain=ch23)
    # (conv_3 reads the first 256 columns of the fm_2 | fm_3 planes, whose per-block magnitudes cover fm_2 alone at this point)
    hs_layer(cv[3], v1, fm2, lambda: feat_graph("conv_3.rf", fm2, k1, v1),
             graphs.get("conv_3.orl_xyz", lambda: xyz_graph(1, v1, k1)), fm3, cv[3]["scale"], cv[3]["shift"], "relu",
             fmap_p=pl.get("fm23"), out_p=pl.get("fm23"), out_col0=256, amax_ws=pl.get("amax_g3"), tickets=pl.get("tick3"),
             proj9=(ch23 or {}).get("proj9"))
    v2, fp2 = ops.pool(v1, fm3, graphs.get("pool_2.xyz", lambda: xyz_graph(1, v1, k1)), s2, kpool=4, planes=pl.get("fp2"))

    P1 = P1_join = None
    if factored and getattr(pk, "fact", None) is not None:
        # the level-1 coarse product W_1 x [fm_2 | fm_3] of the factored wide layers (coarse_products) needs nothing beyond conv_3:
        # on a side stream it runs beside conv_4, the two nearest-point searches, the row sort and the gather -- 160 us of small
        # launches that leave most CUs idle -- instead of in front of conv_5 on the critical path
        f = pk.fact
        p1_fn = lambda: _coarse_product(fm23.reshape(-1, 512), f, "Wb", pl.get("fm23"))
        if not COARSE_SIDE:
            pass                                   # coarse_products computes it in line
        elif BRANCH_STREAMS:
            P1, P1_join = _beside(dev, p1_fn, tag="coarse")
            if not torch.cuda.is_current_stream_capturing():
                fm23.record_stream(_side_stream(dev, (SIDE_TAG, "coarse")))
        else:
            P1 = p1_fn()

    k2 = min(kmax, N2 // 8)
    fm4 = torch.empty(B, N2, 512, device=dev, dtype=torch.float32)
    hs_layer(cv[4], v2, fp2, lambda: feat_graph("conv_4.rf", fp2, k2, v2),
             lambda: graphs.get("conv_4.orl_xyz", lambda: xyz_graph(2, v2, k2)), fm4, fmap_p=pl.get("fp2"), out_p=pl.get("fm4"), amax_ws=pl.get("amax_g4"), tickets=pl.get("tick4"))

    if graphs.inject or graphs.record is not None:
        near1 = graphs.get("up_1", lambda: ops.nn1(xyz, v1)).view(B, N)
        near2 = graphs.get("up_2", lambda: ops.nn1(xyz, v2)).view(B, N)
        tail_done = False
    else:
        # both look-ups in one launch (same results), which also writes the buffer's tail columns when they sit behind fm_1
        near1, near2 = ops.nn1_pair(xyz, v1, v2, tail=(obj_id.reshape(-1).float().contiguous(), feat, 256, n_cls) if factored else None)
        tail_done = factored
    inter = dict(fm_2=fm2, fm_3=fm3, fm_4=fm4, v_pool_1=v1, v_pool_2=v2)
    if factored:
        if not tail_done:
            ops.fill_tail(obj_id.reshape(-1).float(), xyz, feat, 256, n_cls)
        # Everything downstream of the concat buffer in eval mode is a per-point layer followed by a max over the object's
        # points, so the order of an object's rows is free: put the points that share their nearest coarse points next to each
        # other.  A 64-row wave tile of the fine GEMM then fetches ~4 + ~16 distinct coarse rows instead of 128 scattered ones
        # (the gathered rows are 18 KB each; unsorted they stream 1.1 GB through the fabric per forward).
        # (one launch: torch.argsort(near2 * N1 + near1, stable=True), the two gathers and the offsets, ops.sort_by_parent)
        order32, order, near1, near2 = ops.sort_by_parent(near1.contiguous(), near2.contiguous(), N1, N2)
        fine = torch.empty_like(feat)
        if pl:
            ops.planes_gather(feat, order32, fine, FINE_K, pl["fine"])      # the sorted rows and their planes in one pass
        else:
            ops.gather_rows(feat, order32, fine)
        inter.update(fm23=fm23, near1=near1, near2=near2, order=order, P1=P1, P1_join=P1_join, planes=pl)
        return fine, inter
    ops.gather_rows(fm2, near1, feat[:, :, 256:512])
    ops.gather_rows(fm3, near1, feat[:, :, 512:768])
    ops.gather_rows(fm4, near2, feat[:, :, 768:1280])
    ops.fill_tail(obj_id.reshape(-1).float(), xyz, feat, 1280, n_cls)
    return feat, inter


class Arena(object):
    """Everything one eval forward needs zero-initialised -- the order-preserving max keys of conv_5 and of the heads' conv2, the two
    fp16-range flags, the zero-padded topology back-projection -- carved out of ONE buffer zeroed by ONE fill (round 2: five
    fills of 4-5 us each per forward)."""

    def __init__(self, B, dev, N=0, zeroed=True):
        n5, n2, nb = B * 1024, 3 * B * 256, B * FEAT_LD
        # (round 4) + the per-32-row-block magnitudes of the activations that travel as fp16 planes (Arena.planes): five tensors
        # of B*N rows, two of B*N/4, two of B*N/16, and the five graph-convolution outputs (the layers' last GEMM operands)
        N1 = int(N / 4)
        blk = lambda rows: (rows + 31) // 32
        na = 7 * blk(B * N) + 4 * blk(B * N1) + 3 * blk(B * int(N1 / 4)) if N else 0
        na += 5 * B if N else 0                  # + the five graph-convolution layers' tickets (ops.orl_rowbias, one-launch form)
        # zeroed=False: the caller clears `buf` before the first use (posenet_forward: in the centring launch, ops.center(zero=))
        buf = (torch.zeros if zeroed else torch.empty)(n5 + n2 + 8 + nb + na, device=dev, dtype=torch.int32)
        self.buf = buf
        self.keys5 = buf[:n5].view(B, 1024)
        self.keys2 = buf[n5:n5 + n2].view(3, B, 256)
        self.over5 = buf[n5 + n2:n5 + n2 + 1]
        self.over2 = buf[n5 + n2 + 4:n5 + n2 + 5]
        self.dec_flag = buf[n5 + n2 + 6:n5 + n2 + 7]              # the planes-only decoder chain's range flag
        self.chain_flags = (buf[n5 + n2 + 1:n5 + n2 + 2], buf[n5 + n2 + 2:n5 + n2 + 3])      # ops.hs_chain's (conv_0 -> conv_1, conv_2 -> conv_3)
        self.back = buf[n5 + n2 + 8:n5 + n2 + 8 + nb].view(torch.float32).view(B, FEAT_LD)
        self.amax = buf[n5 + n2 + 8 + nb:]
        self._pl = None
        self._repair = None

    def repair_scratch(self, floats):
        """One fp32 scratch for the forward's fp16-range REPAIR launches (the fused heads' two-launch form, the planes-only decoder's
        fp32 chain): they are predicated on device flags and normally return at once, so their activations are never touched --
        and, in a forward without side branches, they run one after the other on one stream: both take views of the same
        buffer (a B = 32 capture keeps 404 MB for them instead of 575).  With side branches the two run concurrently and the
        second caller gets memory of its own."""
        if BRANCH_STREAMS:
            return torch.empty(floats, device=self.buf.device, dtype=torch.float32)
        if self._repair is None or self._repair.numel() < floats:
            self._repair = torch.empty(floats, device=self.buf.device, dtype=torch.float32)
        return self._repair[:floats]

    def planes(self, B, N, N1, N2):
        """the forward's plane buffers (uninitialised; every chunk a consumer reads is written by a producer first) over the
        arena's zeroed magnitude words"""
        if self._pl is None:
            dev = self.amax.device
            spec = (("fm0", B * N, 128), ("fine", B * N, FINE_K), ("d1", B * N, 512), ("d2", B * N, 512), ("d3", B * N, 256),
                    ("fp1", B * N1, 128), ("fm23", B * N1, 512), ("fp2", B * N2, 256), ("fm4", B * N2, 512))
            need = sum((rows + 31) // 32 for _, rows, _ in spec) + 2 * ((B * N + 31) // 32) + 2 * ((B * N1 + 31) // 32) + (B * N2 + 31) // 32
            need += 5 * B
            if self.amax.numel() < need:
                raise RuntimeError("Arena built without room for the planes' magnitudes")
            o, self._pl = 0, {}
            for name, rows, K in spec:
                nb = (rows + 31) // 32
                self._pl[name] = ops.Planes(rows, K, dev, amax_buf=self.amax[o:o + nb])
                o += nb
            # magnitude words of the graph convolutions' outputs (their planes are allocated by the layers)
            for name, rows in (("amax_g0", B * N), ("amax_g1", B * N), ("amax_g2", B * N1), ("amax_g3", B * N1), ("amax_g4", B * N2)):
                nb = (rows + 31) // 32
                self._pl[name] = self.amax[o:o + nb]
                o += nb
            for i in range(5):                     # per layer and object: zero here, handed back as zero by the kernel
                self._pl["tick%d" % i] = self.amax[o:o + B]
                o += B
        return self._pl


def ph_tail(ph, keys, B, dev, back=None, rb_w=None):
    """PH_Predictor after the max over points (FaceRecon.py:145-165): keys = colmax keys of conv_5.  Returns h1, h2, back, rb:
    back = pi1_1 + pi2_1 (B, FEAT_LD) -- or, with rb_w = Packed.rb_w, back None and rb (B, 512) the decoder's per-object bias
    W0 back computed straight from [pi1 | pi2]."""
    if back is None and rb_w is None:
        back = torch.zeros(B, FEAT_LD, device=dev, dtype=torch.float32)
    l1, (w23, b23) = ph["l1"], ph["l23"]
    if PH_TAIL_FUSED and B <= 32 and keys.is_contiguous() and l1.shape[1] == 2 * keys.shape[1]:
        # (round 4) the vector layers read the max keys as they lie -- decoded on load, cat((max, max), 1) as a wrapped column
        # index -- and the sigmoid leaves from the epilogue of the layer it follows: three launches instead of five
        fa = torch.empty(B, l1.shape[0], device=dev, dtype=torch.float32)
        ops.gemm(keys, l1, fa, M=B, N=l1.shape[0], K=l1.shape[1], lda=keys.shape[1], ldw=l1.shape[1], ldc=l1.shape[0],
                 scale=ph["bn5"][0], shift=ph["bn5"][1], act=1, slope=0.2, a_keys=True, a_wrap=keys.shape[1])
        pi = torch.empty(B, w23.shape[0], device=dev, dtype=torch.float32)
        h = torch.empty_like(pi)
        ops.gemm(fa, w23, pi, M=B, N=w23.shape[0], K=w23.shape[1], lda=fa.shape[1], ldw=w23.shape[1], ldc=w23.shape[0], bias=b23,
                 c_sigmoid=h)
    else:
        g = ops.colmax_decode(keys, out2=True)                                       # cat((max, max), 1)
        fa = ops.linear_rows(g, l1, scale=ph["bn5"][0], shift=ph["bn5"][1], act=1, slope=0.2)
        pi = ops.linear_rows(fa, w23, bias=b23)                                      # (B, 5000) = [pi1 | pi2]
        h = ops.sigmoid(pi)
    rb = None
    if rb_w is not None:
        back, rb = None, ops.linear_rows(pi, rb_w[0], bias=rb_w[1])
    else:
        ops.linear_rows(pi, ph["l45"][0], bias=ph["l45"][1], out=back[:, :FEAT_C])   # pi1_1 + pi2_1
    nc = ph["n_code"]
    return h[:, :nc], h[:, nc:], back, rb


def ph_forward(pk, feat, N):
    """PH_Predictor.forward (FaceRecon.py:139-167) -> h1, h2 (B,2500) and back = pi1_1 + pi2_1 (B,FEAT_LD)."""
    B = feat.shape[0]
    ph = pk.ph
    keys = torch.zeros(B, 1024, device=feat.device, dtype=torch.int32)
    ops.linear_rows(feat, ph["w5"], scale=ph["bn5c"][0], shift=ph["bn5c"][1], act=1, slope=0.2, want_out=False,
                    colmax_keys=keys, rows_per_obj=N, k_alg=FEAT_C)
    return ph_tail(ph, keys, B, feat.device)[:3]


def wide_gemm(pk, feat, N):
    """conv_5 + the three head conv1 in one GEMM over `feat` (N_out = 4096).
    Returns (keys5 (B,1024) colmax keys of conv_5, H (B*N, 3072) head activations)."""
    B = feat.shape[0]
    dev = feat.device
    w = pk.wide
    M = B * N
    keys5 = torch.zeros(B, 1024, device=dev, dtype=torch.int32)
    H = torch.empty(M, 3072, device=dev, dtype=torch.float32)
    ops.gemm(feat, w["W"], H, M=M, N=4096, K=FEAT_LD, lda=FEAT_LD, ldw=FEAT_LD, ldc=3072, bias=w["bias"],
             scale=w["scale"], shift=w["shift"], act=1, slope_vec=w["slope"], colmax_keys=keys5, cm_cols=1024,
             c_col0=1024, rows_per_obj=N, k_alg=w["k_alg"], w_split=w["Ws"])
    return keys5, H


def coarse_products(pk, inter, heads_only=False):
    """W_1 x [fm_2 | fm_3] and W_2 x fm_4 for the 4096 columns of the wide layer and the 512 of the decoder's first conv, per
    coarse point: (B*N1, 4608), (B*N2, 4608).  heads_only (EVAL_OUTPUTS_ONLY): only the three heads' 3072 columns are computed
    (the buffers keep their width, so the consumers' column offsets stay)."""
    f = pk.fact
    if heads_only:
        out = []
        for src, W, Ws in ((inter["fm23"], f["Wb"], f["Wb_s"]), (inter["fm_4"], f["Wc"], f["Wc_s"])):
            src = src.reshape(-1, 512)
            P = torch.empty(src.shape[0], W.shape[0], device=src.device, dtype=torch.float32)
            ops.linear_rows(src, W[1024:4096], w_split=ops.split_rows(Ws, 1024, 4096), out=P[:, 1024:4096], flops_ref=0)
            out.append(P)
        return out[0], out[1]
    P1 = inter.get("P1")
    pl = inter.get("planes") or {}
    if P1 is None:
        P1 = _coarse_product(inter["fm23"].reshape(-1, 512), f, "Wb", pl.get("fm23"))
    elif inter.get("P1_join") is not None:        # computed beside conv_4 (encoder_forward): join before the first consumer
        torch.cuda.current_stream(P1.device).wait_event(inter["P1_join"])
    return P1, _coarse_product(inter["fm_4"].reshape(-1, 512), f, "Wc", pl.get("fm4"))


def _coarse_product(x, f, key, x_planes):
    """x (rows, 512) times the stacked coarse weight f[key]: on the projection kernel (round 5: the operand's fragments resident for
    all 4608 columns) when the operand travels as planes and the launch would otherwise take the tile kernels -- whose bits it computes"""
    W = f[key]
    if PROJ_KERNEL and x_planes is not None and f.get(key + "_u") is not None and ops._routes_to_big_tile(x.shape[0], W.shape[0], 1, True):
        return ops.proj_planes(x_planes, f[key + "_u"], None, x, W, flops_ref=0)
    return ops.linear_rows(x, W, w_split=f[key + "_s"], flops_ref=0, a_planes=x_planes, w_planes=f.get(key + "_p"))


def wide_gemm_factored(pk, fine, inter, P1, P2, N, arena=None, heads_only=False):
    """wide_gemm over the fine buffer with the coarse products fetched by the epilogue (same outputs).  heads_only: conv_5 (the PH
    predictor's input) is not computed and keys5 is None."""
    B = fine.shape[0]
    dev = fine.device
    w, f = pk.wide, pk.fact
    M = B * N
    if arena is None:
        arena = Arena(B, dev)
    if HEADS_FUSED and f["w2p"] is not None:
        # conv_5 (N = 1024, only its max over points is used) on the light fused kernel, the heads on theirs; the tile-kernel form
        # of conv_5 follows predicated on the range flag (it normally returns at once)
        light = P1.numel() < 2 ** 30 and P2.numel() < 2 ** 30      # the light kernel addresses the coarse products with 32-bit byte offsets
        over5 = None
        keys5 = None if heads_only else arena.keys5
        if heads_only:
            pass
        elif light:
            keys5, over5 = ops.conv_max_fused(fine.view(M, -1), FINE_K, f["Wa_cp"], P1, inter["near1"], P2, inter["near2"],
                                              w["bias"][:1024], w["scale"][:1024], w["shift"][:1024], 0.2, B, N, k_alg=w["k_alg"],
                                              keys=arena.keys5, overflow=arena.over5, fine_planes=(inter.get("planes") or {}).get("fine"))
        if not heads_only:
            ops.gemm(fine, f["Wa"], None, M=M, N=1024, K=FINE_K, lda=FINE_LD, ldw=FINE_LD, ldc=0, bias=w["bias"],
                     scale=w["scale"], shift=w["shift"], act=1, slope_vec=w["slope"], colmax_keys=keys5, cm_cols=1024,
                     rows_per_obj=N, w_split=f["Wa_s"], gather1=(P1, P1.shape[1], inter["near1"]),
                     gather2=(P2, P2.shape[1], inter["near2"]), flops_ref=0 if light else 2.0 * M * 1024 * w["k_alg"], pred=over5)
        # (returned as a thunk: the caller forks the PH / decoder branch, which needs only keys5, before the long heads kernel)
        def heads():
            # The fused kernel's grid is heads x 128-point workgroups, one per CU and round.  When the last round would hold only
            # a few of them (B = 32, N = 1028: 771 = 3 x 256 + 3) the rows behind the last full round go to the two-launch tile
            # form instead -- 128 rows per head here, two launches of ~10 us on a side stream while the fused kernel runs -- and
            # merge into the same keys (atomicMax on order-preserving keys is exact and order-free).
            wg = 3 * ((M + 127) // 128)
            slots = ops._big_tile_threshold() * 2                         # CUs = resident workgroups of the fused kernel
            rows = 0
            if HEADS_TAIL and wg > slots and wg % slots and (wg % slots) * 8 <= slots:
                rows = ((wg // slots) * slots // 3) * 128
            tail = None
            if rows and rows < M:
                def tail_rows():
                    Mt = M - rows
                    Ht = torch.empty(Mt, 3072, device=dev, dtype=torch.float32)
                    ft = fine.view(M, -1)[rows:]
                    ops.gemm(ft, f["Wa"][1024:], Ht, M=Mt, N=3072, K=FINE_K, lda=FINE_LD, ldw=FINE_LD, ldc=3072, bias=w["bias"][1024:],
                             scale=w["scale"][1024:], shift=w["shift"][1024:], act=1, slope=0.0, rows_per_obj=N, w_split=ops.split_rows(f["Wa_s"], 1024),
                             gather1=(P1[:, 1024:], P1.shape[1], inter["near1"].view(-1)[rows:]),
                             gather2=(P2[:, 1024:], P2.shape[1], inter["near2"].view(-1)[rows:]), flops_ref=0, row_base=rows)
                    ops.gemm(Ht, w["W2"], None, M=Mt, N=256, K=1024, lda=3072, ldw=1024, ldc=0, bias=w["b2"], scale=w["scale2"],
                             shift=w["shift2"], act=1, slope=0.0, colmax_keys=arena.keys2, rows_per_obj=N, batch=3,
                             batch_strides=(1024, 256 * 1024, 0, 256, B * 256), w_split=w["W2s"], flops_ref=0, row_base=rows)
                    return Ht
                if BRANCH_STREAMS:
                    Ht, tail = _beside(dev, tail_rows, tag="heads_tail")
                    if not torch.cuda.is_current_stream_capturing():
                        for t in (fine, P1, P2, inter["near1"], inter["near2"], arena.keys2):
                            t.record_stream(_side_stream(dev, (SIDE_TAG, "heads_tail")))
                else:
                    tail_rows()
            keys2, overflow = ops.heads_fused(fine.view(M, -1), FINE_K, f["Wa_hp"], P1[:, 1024:], inter["near1"], P2[:, 1024:],
                                              inter["near2"], f["w2p"],
                                              w["b2"], w["scale2"], w["shift2"], B, N, k_alg=w["k_alg"], keys=arena.keys2,
                                              overflow=arena.over2, rows=rows, fine_planes=(inter.get("planes") or {}).get("fine"))
            if tail is not None:
                torch.cuda.current_stream(dev).wait_event(tail)
            # fp16 range repair, decided on the device: a wave of the fused kernel that met a magnitude beyond fp16's range wrote no
            # keys and raised `overflow`; the two-launch form (whose tiles guard themselves) then supplies every key.  While the
            # flag is 0 -- always, for sane weights -- both launches return at once (tgp_gemm_args.pred).
            # The repair's conv1 activation is (rows, 3072) fp32 -- 404 MB for B = 32 objects of 1028 points, 3.2 GB for 256 -- and
            # sits in every captured forward's pool although the launches normally return at once: the repair therefore walks
            # the batch in chunks of REPAIR_OBJS objects through ONE chunk-sized buffer (tgp_gemm_args.row_base addresses the
            # chunk's objects in the max over points): two predicated launches per chunk.
            Rr = M if B <= 2 * REPAIR_OBJS else REPAIR_OBJS * N       # (up to 2 x REPAIR_OBJS objects: one chunk, two launches)
            H = arena.repair_scratch(Rr * 3072).view(Rr, 3072)
            f2, n1v, n2v = fine.view(M, -1), inter["near1"].view(-1), inter["near2"].view(-1)
            for r0 in range(0, M, Rr):
                Mt = min(Rr, M - r0)
                ops.gemm(f2[r0:], f["Wa"][1024:], H, M=Mt, N=3072, K=FINE_K, lda=FINE_LD, ldw=FINE_LD, ldc=3072, bias=w["bias"][1024:],
                         scale=w["scale"][1024:], shift=w["shift"][1024:], act=1, slope=0.0, rows_per_obj=N,
                         w_split=ops.split_rows(f["Wa_s"], 1024), gather1=(P1[:, 1024:], P1.shape[1], n1v[r0:]),
                         gather2=(P2[:, 1024:], P2.shape[1], n2v[r0:]), flops_ref=0, pred=overflow, row_base=r0)
                ops.gemm(H, w["W2"], None, M=Mt, N=256, K=1024, lda=3072, ldw=1024, ldc=0, bias=w["b2"], scale=w["scale2"],
                         shift=w["shift2"], act=1, slope=0.0, colmax_keys=keys2, rows_per_obj=N, batch=3,
                         batch_strides=(1024, 256 * 1024, 0, 256, B * 256), w_split=w["W2s"], flops_ref=0, pred=overflow, row_base=r0)
            return keys2
        return keys5, heads
    keys5 = arena.keys5
    H = torch.empty(M, 3072, device=dev, dtype=torch.float32)
    ops.gemm(fine, f["Wa"], H, M=M, N=4096, K=FINE_K, lda=FINE_LD, ldw=FINE_LD, ldc=3072, bias=w["bias"],
             scale=w["scale"], shift=w["shift"], act=1, slope_vec=w["slope"], colmax_keys=keys5, cm_cols=1024,
             c_col0=1024, rows_per_obj=N, w_split=f["Wa_s"], gather1=(P1, P1.shape[1], inter["near1"]),
             gather2=(P2, P2.shape[1], inter["near2"]), flops_ref=2.0 * M * 4096 * w["k_alg"])
    return keys5, H


def decoder_forward_factored(pk, fine, inter, P1, P2, back, N, arena=None, rb=None):
    """decoder_forward with the first conv factored like the wide layer (its coarse products are columns 4096.. of P1 / P2).
    rb: W0 back, where the caller has it already (ph_tail(rb_w=...))."""
    w0, b0, sc0, sh0 = pk.dec[0][:4]
    f = pk.fact
    B = fine.shape[0]
    M = B * N
    if rb is None:
        rb = ops.linear_rows(back, w0) if back is not None else None
    # (the light fused kernel in a storing form was measured for this layer: 110 us against 96 us on the tile kernel -- with 512
    # channels a workgroup has 8 channel blocks to amortise its set-up over, and 4-byte stores of 16 points per lane)
    pl = inter.get("planes") or {}
    dev = fine.device
    gk = dict(M=M, N=512, K=FINE_K, lda=FINE_LD, ldw=FINE_LD, ldc=512, bias=b0, rowbias=rb, rows_per_obj=N, scale=sc0, shift=sh0, act=1,
              w_split=f["dec_a_s"], gather1=(P1[:, 4096:], P1.shape[1], inter["near1"]),
              gather2=(P2[:, 4096:], P2.shape[1], inter["near2"]))
    chain = pl.get("fine") is not None and f.get("dec_a_p") is not None and all(d[5] is not None for d in pk.dec[1:])
    widths = [512] + [d[0].shape[0] for d in pk.dec[1:]]                  # 512, 512, 256, 128
    if chain and DEC_PLANES_ONLY and arena is not None and all(ops._routes_to_big_tile(M, n, 1, True) for n in widths):
        # The chain fine -> 512 -> 512 -> 256 -> 128 with its three inner activations as fp16 planes ONLY: each layer's epilogue
        # writes the next one's operand (4 bytes per element written and read, instead of 8 + 4).  A tile whose magnitude words lie
        # outside fp16's range cannot be recomputed without the fp32 operand: it raises arena.dec_flag, and the fp32 chain follows
        # predicated on the flag (four launches that normally return at once, as after the fused heads kernel).
        flag = arena.dec_flag
        fused = DEC_FUSED and ROWS_OUT and getattr(pk, "dec_units", None) is not None
        l1 = fused and DEC_L1 and f.get("dec_a_hp") is not None and getattr(f["dec_a_s"], "tgp_unscale", None) is None
        if l1:
            # (round 5) the first conv on the conv1 half of the fused heads kernel: its result goes straight to the fused decoder
            # kernel as fragments in accumulator order (no LDS transposition, 257 workgroups instead of 2056)
            xp = ops.dec_l1(pl["fine"], f["dec_a_hp"], P1[:, 4096:], inter["near1"], P2[:, 4096:], inter["near2"], b0, sc0, sh0, rb, N, pl["d1"],
                            flag, k_alg=FEAT_C)
        else:
            ops.gemm(fine, f["dec_a"], None, flops_ref=2.0 * M * 512 * FEAT_C, a_planes=pl["fine"], w_planes=f["dec_a_p"], c_planes=pl["d1"], **gk)
            xp = pl["d1"]
        x = torch.empty(M, widths[-1], device=dev, dtype=torch.float32)
        if fused:
            # (round 5) 512 -> 512 -> 256 -> 128 -> 3 and the un-sort of the rows as ONE launch: no activation between the layers leaves
            # the registers.  Same flag: a wave whose operand block or sums leave fp16's range raises it and the predicated fp32 chain
            # below -- now with a predicated last step -- rewrites the reconstruction.
            recon = ops.dec_fused(xp, pk.dec_units_p if l1 else pk.dec_units, [d[1:4] for d in pk.dec[1:]], pk.dec_out[0], pk.dec_out[1],
                                  inter["order"].contiguous(), N, flag).view(B, N, 3)
        for i, ((w, b, sc, sh, ws, wp), nxt) in enumerate(() if fused else zip(pk.dec[1:], ("d2", "d3", None))):
            ops.gemm(None, w, x if nxt is None else None, M=M, N=w.shape[0], K=w.shape[1], lda=0, ldw=w.shape[1], ldc=w.shape[0], bias=b,
                     scale=sc, shift=sh, act=1, w_split=ws, a_planes=xp, w_planes=wp, c_planes=pl.get(nxt) if nxt else None, range_flag=flag)
            xp = pl.get(nxt) if nxt else None
        # the repair chain's activations (untouched while the flag is 0): 512 | 512 | 256 columns out of the forward's repair scratch
        rs = arena.repair_scratch(M * sum(widths[:-1]))
        offs = [0]
        for n in widths[:-1]:
            offs.append(offs[-1] + M * n)
        y = rs[offs[0]:offs[1]].view(M, 512)
        ops.gemm(fine, f["dec_a"], y, flops_ref=0, pred=flag, **gk)
        for li, (w, b, sc, sh, ws, _wp) in enumerate(pk.dec[1:]):
            out = x if w.shape[0] == widths[-1] else rs[offs[li + 1]:offs[li + 2]].view(M, w.shape[0])
            ops.gemm(y, w, out, M=M, N=w.shape[0], K=w.shape[1], lda=w.shape[1], ldw=w.shape[1], ldc=w.shape[0], bias=b, scale=sc,
                     shift=sh, act=1, w_split=ws, flops_ref=0, pred=flag)
            y = out
        x = x.view(B, N, -1)
        if fused:
            return ops.rows_out(x, pk.dec_out[0], pk.dec_out[1], inter["order"].contiguous(), out=recon, pred=flag)
    else:
        x = torch.empty(B, N, 512, device=dev, dtype=torch.float32)
        # the chain on the pre-split kernel: each layer's epilogue leaves the next one's operand planes (the fp32 copies are written
        # too: a tile outside fp16's range recomputes from them, csrc/gemm_pp.hip)
        ops.gemm(fine, f["dec_a"], x, flops_ref=2.0 * M * 512 * FEAT_C, a_planes=pl.get("fine"), w_planes=f.get("dec_a_p"),
                 c_planes=pl.get("d1") if chain else None, **gk)
        xp = pl.get("d1") if chain else None
        for (w, b, sc, sh, ws, wp), nxt in zip(pk.dec[1:], ("d2", "d3", None)):
            x = ops.linear_rows(x, w, bias=b, scale=sc, shift=sh, act=1, w_split=ws, a_planes=xp if wp is not None else None, w_planes=wp,
                                c_planes=pl.get(nxt) if (nxt and xp is not None and wp is not None) else None)
            xp = pl.get(nxt) if (nxt and xp is not None and wp is not None) else None
    # rows are in the sorted order of encoder_forward(factored=True): the (B,N,3) result goes back in point order
    if ROWS_OUT and pk.dec_out[0].shape[0] <= 4 and pk.dec_out[0].shape[1] % 4 == 0:
        return ops.rows_out(x, pk.dec_out[0], pk.dec_out[1], inter["order"].contiguous())
    recon = ops.linear_rows(x, pk.dec_out[0], bias=pk.dec_out[1])
    return torch.empty_like(recon).scatter_(1, inter["order"].unsqueeze(-1).expand(-1, -1, 3), recon)


PH_RB_COMPOSED = os.environ.get("TGP_PH_RB_COMPOSED", "1") != "0"   # decoder's per-object bias from [pi1 | pi2] through W0 L45
KNN_DIRS = os.environ.get("TGP_KNN_DIRS", "1") != "0"        # feature kNN leaves the unit neighbour directions beside its lists
PH_TAIL_FUSED = os.environ.get("TGP_PH_TAIL_FUSED", "1") != "0"   # PH predictor's vector layers: key decode and sigmoid inside them
ROWS_OUT = os.environ.get("TGP_ROWS_OUT", "1") != "0"        # the decoder's last conv and the un-sort of its rows as one launch
POSE_TAIL = os.environ.get("TGP_POSE_TAIL", "1") != "0"      # conv3, conv4 and the output formulas of the heads as one launch


def head_chain(pk, H, B, N, mean=None):
    """The three heads after conv1: conv2 (+BN, ReLU, max over points) as one batched launch, then conv3 / conv4
    batched over the heads.  Returns [green (B,4), red (B,4), ts (B,6)]; with `mean` (the clouds' centres) the six pose outputs
    of PoseNet9D.py:57-66 (p_green_R, p_red_R, f_green_R, f_red_R, Pred_T, Pred_s) instead."""
    dev = H.device
    w = pk.wide
    M = B * N
    if H.dtype == torch.int32:      # wide_gemm_factored ran the fused heads kernel: H is already conv2's pooled keys (3, B, 256)
        keys2 = H
    else:
        keys2 = torch.zeros(3, B, 256, device=dev, dtype=torch.int32)
        ops.gemm(H, w["W2"], None, M=M, N=256, K=1024, lda=3072, ldw=1024, ldc=0, bias=w["b2"], scale=w["scale2"],
                 shift=w["shift2"], act=1, slope=0.0, colmax_keys=keys2, rows_per_obj=N, batch=3,
                 batch_strides=(1024, 256 * 1024, 0, 256, B * 256), w_split=w["W2s"])
    if mean is not None and POSE_TAIL and w.get("W3t") is not None:
        return ops.pose_tail(keys2.contiguous(), w["W3t"], w["b3"], w["scale3"], w["shift3"], w["W4"], w["b4"], mean)
    pooled = ops.colmax_decode(keys2.view(3 * B, 256))                          # (3B, 256)
    # conv3 (+BN, ReLU), dropout(eval) = identity, conv4: two batched launches for the three heads
    x3 = torch.empty(3, B, 256, device=dev, dtype=torch.float32)
    ops.gemm(pooled, w["W3"], x3, M=B, N=256, K=256, lda=256, ldw=256, ldc=256, bias=w["b3"], scale=w["scale3"],
             shift=w["shift3"], act=1, slope=0.0, batch=3, batch_strides=(B * 256, 256 * 256, B * 256, 256, 0))
    o4 = torch.empty(3, B, 8, device=dev, dtype=torch.float32)
    ops.gemm(x3, w["W4"], o4, M=B, N=8, K=256, lda=256, ldw=256, ldc=8, bias=w["b4"], batch=3,
             batch_strides=(B * 256, 8 * 256, B * 8, 8, 0))
    green, red, ts = [o4[i, :, : w["n_out"][i]] for i in range(3)]  # views of the (3, B, 8) buffer: tgp_head_post takes row strides
    return (green, red, ts) if mean is None else ops.head_post(green, red, ts, mean)


def wide_forward(pk, feat, N):
    keys5, H = wide_gemm(pk, feat, N)
    return keys5, head_chain(pk, H, feat.shape[0], N)


def head_tail(hd, pooled):
    """conv3 (+BN, ReLU), dropout(eval) = identity, conv4 on the pooled (B,256) vector."""
    w, b, sc, sh = hd["c3"]
    x = ops.linear_rows(pooled, w, bias=b, scale=sc, shift=sh, act=1)
    return ops.linear_rows(x, hd["c4"][0], bias=hd["c4"][1])


def decoder_forward(pk, feat, back, N, rb=None):
    """Face_Dec.forward on feat + back (FaceRecon.py:165,112-117) -> recon (B,N,3).

    conv(feat + back) = conv(feat) + W @ back: the broadcast add of the topology code becomes a
    per-object bias of the first GEMM, so (B,1286,N) feat_ph is never materialised."""
    B = feat.shape[0]
    w0, b0, sc0, sh0, ws0 = pk.dec[0][:5]
    x = feat
    if rb is None:
        rb = ops.linear_rows(back, w0) if back is not None else None
    x = ops.linear_rows(x, w0, bias=b0, rowbias=rb, rows_per_obj=N, scale=sc0, shift=sh0, act=1, k_alg=FEAT_C, w_split=ws0)
    for w, b, sc, sh, ws, _ in pk.dec[1:]:
        x = ops.linear_rows(x, w, bias=b, scale=sc, shift=sh, act=1, w_split=ws)
    return ops.linear_rows(x, pk.dec_out[0], bias=pk.dec_out[1])


def head_forward(hd, feat, N):
    """Rot_green / Rot_red / Pose_Ts (PoseR.py:26-39, PoseTs.py:31-45) -> (B, 4|6)."""
    B = feat.shape[0]
    w, b, sc, sh = hd["c1"]
    x = ops.linear_rows(feat, w, bias=b, scale=sc, shift=sh, act=1, k_alg=hd["k_alg"])
    w, b, sc, sh = hd["c2"]
    keys = torch.zeros(B, w.shape[0], device=feat.device, dtype=torch.int32)
    ops.linear_rows(x, w, bias=b, scale=sc, shift=sh, act=1, want_out=False, colmax_keys=keys, rows_per_obj=N)
    x = ops.colmax_decode(keys)
    w, b, sc, sh = hd["c3"]
    x = ops.linear_rows(x, w, bias=b, scale=sc, shift=sh, act=1)
    return ops.linear_rows(x, hd["c4"][0], bias=hd["c4"][1])


def draw_sample_idx(N):
    """The two ``torch.randperm`` draws of Face_Enc.forward, from the global CPU generator in the
    reference's order (pool_1 then pool_2; gcn3d.py:241-242)."""
    i1 = torch.randperm(N)[: int(N / 4)]
    i2 = torch.randperm(i1.numel())[: int(i1.numel() / 4)]
    return i1, i2


def posenet_forward(pk, points, obj_id, train_keys, sample_idx=None, inject=None, record=None, kmax=20, n_cls=6, probe=None,
                    outputs_only=None, proj=None):
    """PoseNet9D(only_encoder=False).forward in eval mode (PoseNet9D.py:46-91).
    probe (tests): a dict that receives what the six-key eval result does not return -- recon (with the cloud's mean added), h1,
    h2, feat_global -- so that the factored / fused eval paths' decoder and PH branch can be compared with the concat path's."""
    if points.dim() != 3 or points.shape[2] != 3:
        raise ValueError("points must be (B,N,3)")
    B, N, _ = points.shape
    if N // 4 // 4 // 8 < 1:
        raise ValueError("need at least 128 points per object (k = min(20, n // 8) must be >= 1 at every level)")
    if sample_idx is None:
        sample_idx = draw_sample_idx(N)
    points = points.contiguous().float()
    graphs = Graphs(points.device, inject, record)
    factored = FACTORED and not train_keys          # the concat buffer is an output only with the training keys
    # (zeroed on this stream before any branch forks: by the centring launch itself)
    arena = Arena(B, points.device, N, zeroed=False) if factored else None
    xyz, mean = ops.center(points, zero=arena.buf if arena is not None else None)
    feat, inter = encoder_forward(pk, xyz, obj_id.to(points.device), sample_idx, graphs, kmax, n_cls, factored=factored, arena=arena)
    # EVAL_OUTPUTS_ONLY: the six-key eval dict (PoseNet9D.py:85-90) needs neither the PH predictor nor the decoder -- the reference
    # computes both and drops them.  Off by default: the bench's headline is the full forward (SURVEY 8d's algorithmic figures).
    # outputs_only: this call's setting (PoseNet9D.forward hands its caller's down); None = the process-wide default
    heads_only = ((EVAL_OUTPUTS_ONLY if outputs_only is None else bool(outputs_only)) and factored and probe is None and HEADS_FUSED
                  and getattr(pk, "fact", None) is not None
                  and pk.fact["w2p"] is not None)
    if heads_only:
        P1, P2 = coarse_products(pk, inter, heads_only=True)
        _, H = wide_gemm_factored(pk, feat, inter, P1, P2, N, arena, heads_only=True)
        pg, pr, fg, fr, pT, ps = head_chain(pk, H(), B, N, mean)
        return dict(p_green_R=pg, p_red_R=pr, f_green_R=fg, f_red_R=fr, Pred_T=pT, Pred_s=ps)
    if factored:
        P1, P2 = coarse_products(pk, inter)
        wide = lambda: wide_gemm_factored(pk, feat, inter, P1, P2, N, arena)
        decode = lambda back, rb: decoder_forward_factored(pk, feat, inter, P1, P2, back, N, arena, rb)
    else:
        P1 = P2 = None
        wide = lambda: wide_gemm(pk, feat, N)
        decode = lambda back, rb: decoder_forward(pk, feat, back, N, rb)
    rb_w = getattr(pk, "rb_w", None) if PH_RB_COMPOSED else None
    if BRANCH_STREAMS:
        # after the fused wide GEMM the head chain (conv2 -> max -> conv3 -> conv4) and the PH tail -> decoder chain are
        # independent: the second runs on a side stream and fills the tail rounds / skinny launches of the first
        cur = torch.cuda.current_stream(points.device)
        side = _side_stream(points.device)
        keys5, H = wide()
        fork = torch.cuda.Event()
        fork.record(cur)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            h1, h2, back, rb = ph_tail(pk.ph, keys5, B, points.device, arena.back if arena is not None else None, rb_w)
            recon = decode(back, rb)
            join = torch.cuda.Event()
            join.record(side)
        if callable(H):
            H = H()                                           # the fused heads kernel, beside the side branch
        if not torch.cuda.is_current_stream_capturing():      # a captured graph owns its pool: nothing to protect
            for t in (keys5, H, feat, h1, h2, recon) + ((back,) if back is not None else (rb,)) + ((P1, P2, inter["near1"], inter["near2"]) if factored else ()):
                t.record_stream(side)
        pg, pr, fg, fr, pT, ps = head_chain(pk, H, B, N, mean)
        cur.wait_event(join)
    else:
        keys5, H = wide()
        if callable(H):
            H = H()
        pg, pr, fg, fr, pT, ps = head_chain(pk, H, B, N, mean)
        h1, h2, back, rb = ph_tail(pk.ph, keys5, B, points.device, arena.back if arena is not None else None, rb_w)
        recon = decode(back, rb)
    out = dict()
    if probe is not None:
        probe.update(recon=recon + mean.view(B, 1, 3), h1=h1, h2=h2, keys5=keys5)
    if train_keys:
        out["recon"] = ops.add_mean_(recon, mean)
    out.update(p_green_R=pg, p_red_R=pr, f_green_R=fg, f_red_R=fr, Pred_T=pT, Pred_s=ps)
    if train_keys:
        out["h1"], out["h2"] = h1, h2
        out["feat"] = feat[:, :, :FEAT_C]
        # (proj: enable_proj=True -- (operands, state dict); PoseNet9D.py:49-50)
        out["feat_global"] = ops.colmax(feat[:, :, :FEAT_C]) if proj is None else proj_feat_global(proj[0], feat, sd=proj[1])
    return out


def encoder_only_forward(pk, points, obj_id, sample_idx=None, inject=None, record=None, kmax=20, n_cls=6, proj=None):
    """PoseNet9D(only_encoder=True).forward (PoseNet9D.py:35-45): encoder + decoder without the PH branch."""
    B, N, _ = points.shape
    if sample_idx is None:
        sample_idx = draw_sample_idx(N)
    xyz, mean = ops.center(points.contiguous().float())
    graphs = Graphs(points.device, inject, record, prefix="face_enc.encoder.")
    feat, _ = encoder_forward(pk, xyz, obj_id.to(points.device), sample_idx, graphs, kmax, n_cls)
    recon = decoder_forward(pk, feat, None, N)
    fg = ops.colmax(feat[:, :, :FEAT_C]) if proj is None else proj_feat_global(proj[0], feat, sd=proj[1])
    return dict(feat_global=fg, recon=recon)


# =====================================================================================================
# Training-mode forward (module.training == True) without autograd: BatchNorm uses batch statistics and updates its
# running statistics, dropout is active.  This is what the reference's net2 runs under torch.no_grad() every step
# (trainer/RL_TDA.py:117-118); with gradients recorded (net1) the differentiable path of autograd.py runs instead.
# The dense layers run the same GEMM kernels without the folded scale/shift; tgp_bn_stats / tgp_bn_apply then
# normalise in place (two deterministic reduction passes + one apply pass over the activation).
# =====================================================================================================
class TrainBN(object):
    def __init__(self, sd, momentum=0.1, update_running=True):
        self.sd, self.momentum, self.update = sd, momentum, update_running

    def _update(self, name, mean, var, rows):
        if not self.update:
            return
        rm, rv = self.sd[name + ".running_mean"], self.sd[name + ".running_var"]
        m = self.momentum
        rm.mul_(1 - m).add_(mean, alpha=m)
        rv.mul_(1 - m).add_(var, alpha=m * rows / max(rows - 1, 1))     # unbiased estimate, as nn.BatchNorm1d
        self.sd[name + ".num_batches_tracked"].add_(1)

    def __call__(self, name, x, act=0, slope=0.0, **kw):
        running = None
        if self.update:      # the statistics kernel moves the module's buffers itself (tgp_bn_stats_running)
            running = (self.sd[name + ".running_mean"], self.sd[name + ".running_var"], self.momentum,
                       self.sd.get(name + ".num_batches_tracked"))
        out, mean, var = ops.bn_train(x, self.sd[name + ".weight"], self.sd[name + ".bias"], BN_EPS, act, slope, running=running, **kw)
        return out

    def multi(self, names, x, act, slope_vec, **kw):
        """one normalisation over columns that belong to several BatchNorm modules (the fused wide GEMM)"""
        gamma = torch.cat([self.sd[n + ".weight"] for n in names])
        beta = torch.cat([self.sd[n + ".bias"] for n in names])
        out, mean, var = ops.bn_train(x, gamma, beta, BN_EPS, act, 0.0, slope_vec=slope_vec, **kw)
        rows, o = math.prod(x.shape[:-1]), 0
        for n in names:
            c = self.sd[n + ".weight"].numel()
            self._update(n, mean[o:o + c], var[o:o + c], rows)
            o += c
        return out


def encoder_forward_train(pk, bn, points_c, obj_id, sample_idx, graphs, kmax=20, n_cls=6):
    """Face_Enc.forward with batch-statistics BatchNorm (bn1..bn3)."""
    e = pk.face + "encoder."
    dev = points_c.device
    B, N, _ = points_c.shape
    xyz = points_c
    feat = torch.empty(B, N, FEAT_LD, device=dev, dtype=torch.float32)
    N1, N2 = sample_idx[0].numel(), sample_idx[1].numel()
    s12 = _upload_i32(torch.cat([sample_idx[0].reshape(-1), sample_idx[1].reshape(-1)]), dev)
    s1, s2 = s12[:N1], s12[N1:]
    knn0 = {}

    def xyz_graph(level, pts, k):
        if level not in knn0:
            knn0[level] = ops.knn_xyz(pts, k)
        return knn0[level]

    cv = pk.conv
    fm0 = feat[:, :, 0:128]
    surface_layer(cv[0], xyz, graphs.get("conv_0.rf", lambda: xyz_graph(0, xyz, kmax)),
                  graphs.get("conv_0.orl_xyz", lambda: xyz_graph(0, xyz, kmax)), fm0, act="relu")
    fm1 = feat[:, :, 128:256]
    hs_layer(cv[1], xyz, fm0, graphs.get("conv_1.rf", lambda: ops.knn_feat(fm0, kmax)),
                    graphs.get("conv_1.orl_xyz", lambda: xyz_graph(0, xyz, kmax)), fm1)
    bn(e + "bn1", fm1, act=1)
    v1, fp1 = ops.pool(xyz, fm1, graphs.get("pool_1.xyz", lambda: xyz_graph(0, xyz, kmax)), s1, kpool=4)
    k1 = min(kmax, N1 // 8)
    fm2 = torch.empty(B, N1, 256, device=dev, dtype=torch.float32)
    hs_layer(cv[2], v1, fp1, graphs.get("conv_2.rf", lambda: ops.knn_feat(fp1, k1)),
                    graphs.get("conv_2.orl_xyz", lambda: xyz_graph(1, v1, k1)), fm2)
    bn(e + "bn2", fm2, act=1)
    fm3 = torch.empty(B, N1, 256, device=dev, dtype=torch.float32)
    hs_layer(cv[3], v1, fm2, graphs.get("conv_3.rf", lambda: ops.knn_feat(fm2, k1)),
                    graphs.get("conv_3.orl_xyz", lambda: xyz_graph(1, v1, k1)), fm3)
    bn(e + "bn3", fm3, act=1)
    v2, fp2 = ops.pool(v1, fm3, graphs.get("pool_2.xyz", lambda: xyz_graph(1, v1, k1)), s2, kpool=4)
    k2 = min(kmax, N2 // 8)
    fm4 = torch.empty(B, N2, 512, device=dev, dtype=torch.float32)
    hs_layer(cv[4], v2, fp2, graphs.get("conv_4.rf", lambda: ops.knn_feat(fp2, k2)),
             graphs.get("conv_4.orl_xyz", lambda: xyz_graph(2, v2, k2)), fm4)
    if graphs.inject or graphs.record is not None:
        near1 = graphs.get("up_1", lambda: ops.nn1(xyz, v1)).view(B, N)
        near2 = graphs.get("up_2", lambda: ops.nn1(xyz, v2)).view(B, N)
    else:
        near1, near2 = ops.nn1_pair(xyz, v1, v2)          # both look-ups in one launch (same results)
    ops.gather_rows(fm2, near1, feat[:, :, 256:512])
    ops.gather_rows(fm3, near1, feat[:, :, 512:768])
    ops.gather_rows(fm4, near2, feat[:, :, 768:1280])
    ops.fill_tail(obj_id.reshape(-1).float(), xyz, feat, 1280, n_cls)
    return feat


def decoder_forward_train(pk, bn, feat, back, N):
    d = pk.face + "decoder."
    names = (d + "conv1d_block.1", d + "conv1d_block.4", d + "conv1d_block.7", d + "recon_head.1")
    w0, b0, _, _, ws0 = pk.dec[0][:5]
    rb = ops.linear_rows(back, w0) if back is not None else None
    x = ops.linear_rows(feat, w0, bias=b0, rowbias=rb, rows_per_obj=N, w_split=ws0)
    bn(names[0], x, act=1)
    for (w, b, _, _, ws, _p), nm in zip(pk.dec[1:], names[1:]):
        x = ops.linear_rows(x, w, bias=b, w_split=ws)
        bn(nm, x, act=1)
    return ops.linear_rows(x, pk.dec_out[0], bias=pk.dec_out[1])


def posenet_forward_train(pk, sd, points, obj_id, train_keys, sample_idx=None, inject=None, record=None, kmax=20, n_cls=6,
                          dropout_p=(0.5, 0.2), update_running=True, generator=None, proj=None):
    """PoseNet9D(only_encoder=False).forward with module.training == True (no autograd)."""
    B, N, _ = points.shape
    if B < 2:     # bn5 / the heads' bn3 normalise over the B pooled rows; torch refuses a single row the same way
        raise ValueError("Expected more than 1 value per channel when training, got input size [%d, 256]" % B)
    if sample_idx is None:
        sample_idx = draw_sample_idx(N)
    dev = points.device
    points = points.contiguous().float()
    bn = TrainBN(sd, update_running=update_running)
    xyz, mean = ops.center(points)
    graphs = Graphs(dev, inject, record)
    feat = encoder_forward_train(pk, bn, xyz, obj_id.to(dev), sample_idx, graphs, kmax, n_cls)
    M = B * N
    w = pk.wide
    ph = pk.face + "ph_pred."
    # conv_5 | three head conv1 in one GEMM (bias only), then one batch-statistics normalisation over the 4096 columns
    Hall = torch.empty(M, 4096, device=dev, dtype=torch.float32)
    ops.gemm(feat, w["W"], Hall, M=M, N=4096, K=FEAT_LD, lda=FEAT_LD, ldw=FEAT_LD, ldc=4096, bias=w["bias"], w_split=w["Ws"],
             k_alg=w["k_alg"])
    keys5 = torch.zeros(B, 1024, device=dev, dtype=torch.int32)
    bn.multi([ph + "conv_5.1"] + [h + ".bn1" for h in HEAD_ORDER], Hall, 1, w["slope"], colmax_keys=keys5, cm_cols=1024,
             rows_per_obj=N)
    # PH tail
    g = ops.colmax_decode(keys5, out2=True)
    fa = ops.linear_rows(g, pk.ph["l1"])
    bn(ph + "bn5", fa, act=1, slope=0.2)
    fa = ops.dropout(fa, dropout_p[0], generator)
    pi = ops.linear_rows(fa, pk.ph["l23"][0], bias=pk.ph["l23"][1])
    back = torch.zeros(B, FEAT_LD, device=dev, dtype=torch.float32)
    ops.linear_rows(pi, pk.ph["l45"][0], bias=pk.ph["l45"][1], out=back[:, :FEAT_C])
    hcode = ops.sigmoid(pi)
    nc = pk.ph["n_code"]
    recon = decoder_forward_train(pk, bn, feat, back, N)
    # heads: conv2 raw for the three heads in one batched launch, per-head BatchNorm + ReLU + max over points
    X2 = torch.empty(3, M, 256, device=dev, dtype=torch.float32)
    ops.gemm(Hall[:, 1024:], w["W2"], X2, M=M, N=256, K=1024, lda=4096, ldw=1024, ldc=256, bias=w["b2"], batch=3,
             batch_strides=(1024, 256 * 1024, M * 256, 256, 0), w_split=w["W2s"])
    keys2 = torch.zeros(3, B, 256, device=dev, dtype=torch.int32)
    outs = []
    for i, h in enumerate(HEAD_ORDER):
        bn(h + ".bn2", X2[i], act=1, colmax_keys=keys2[i], rows_per_obj=N, want_out=False)
    pooled = ops.colmax_decode(keys2.view(3 * B, 256)).view(3, B, 256)
    for i, h in enumerate(HEAD_ORDER):
        hd = pk.heads[h]
        x = ops.linear_rows(pooled[i], hd["c3"][0], bias=hd["c3"][1])
        bn(h + ".bn3", x, act=1)
        x = ops.dropout(x, dropout_p[1], generator)
        outs.append(ops.linear_rows(x, hd["c4"][0], bias=hd["c4"][1]))
    green, red, ts = outs
    pg, pr, fg, fr, pT, ps = ops.head_post(green, red, ts, mean)
    out = dict()
    if train_keys:
        out["recon"] = ops.add_mean_(recon, mean)
    out.update(p_green_R=pg, p_red_R=pr, f_green_R=fg, f_red_R=fr, Pred_T=pT, Pred_s=ps)
    if train_keys:
        out["h1"], out["h2"] = hcode[:, :nc], hcode[:, nc:]
        out["feat"] = feat[:, :, :FEAT_C]
        out["feat_global"] = ops.colmax(feat[:, :, :FEAT_C]) if proj is None else proj_feat_global(proj[0], feat, bn=bn)
    return out


def encoder_only_forward_train(pk, sd, points, obj_id, sample_idx=None, inject=None, record=None, kmax=20, n_cls=6,
                               update_running=True, proj=None):
    """PoseNet9D(only_encoder=True).forward in training mode (the reference's net2)."""
    B, N, _ = points.shape
    if sample_idx is None:
        sample_idx = draw_sample_idx(N)
    bn = TrainBN(sd, update_running=update_running)
    xyz, mean = ops.center(points.contiguous().float())
    graphs = Graphs(points.device, inject, record, prefix="face_enc.encoder.")
    feat = encoder_forward_train(pk, bn, xyz, obj_id.to(points.device), sample_idx, graphs, kmax, n_cls)
    recon = decoder_forward_train(pk, bn, feat, None, N)
    fg = ops.colmax(feat[:, :, :FEAT_C]) if proj is None else proj_feat_global(proj[0], feat, bn=bn)
    return dict(feat_global=fg, recon=recon)


# =====================================================================================================
# hipGraph replay of the eval-mode forward.  One forward is ~100 kernel launches of 5-1000 us; captured once per
# (B, N, output set) with torch.cuda.graph (hipStreamBeginCapture around the same C-ABI launches) and replayed, the
# launch overhead and the gaps between dependent kernels disappear from the host's and the GPU's timeline.  With
# parts = 2 the batch is captured as two half batches on two forked streams: eval-mode objects are independent
# (SURVEY.md 8e), so the halves' tail rounds, small kernels and epilogues overlap each other inside one graph.
# Inputs are copied into static buffers before a replay; the returned tensors are the graph's static outputs, valid
# until the next replay (the usual contract of graph replay).
# =====================================================================================================
def new_graph():
    """a CUDAGraph that keeps its hipGraph after capture, so that check_capture can take the node census"""
    return torch.cuda.CUDAGraph(keep_graph=True)


def check_capture(graph, what):
    """Refuse a captured graph that holds a MEMSET node, then instantiate it.

    Root cause of the round-2 device fault (scripts/capture_memset_probe.py is the 30-line reproducer): on
    this ROCm stack a hipMemsetAsync captured into a hipGraph is, from the second launch of the graph on, NOT ordered against
    the kernel node in front of it -- half of its words still hold what that kernel wrote.  ATen's multi-block reductions
    (max-with-indices over a long dimension, sums over broadcast dimensions) zero their semaphores exactly that way; inside a
    captured step their semaphore block is recycled pool memory that earlier kernels of the same graph wrote, so from the second
    replay on the reduction's last-block logic does not fire and its outputs keep stale bytes (arg-max "indices" that are really
    fp16 weight planes -> the out-of-bounds scatter of r02a).  Nothing in this library issues a memset; a memset node therefore
    means a torch op of that kind slipped into the captured region, which is refused here instead of being found as a fault."""
    kernels, memcpys, memsets, other = ops.graph_node_counts(graph.raw_cuda_graph())
    if memsets and not os.environ.get("TGP_ALLOW_GRAPH_MEMSET"):
        raise RuntimeError(
            "%s: the captured graph holds %d memset node(s) (and %d kernel nodes).  hipGraph memset nodes are not ordered against "
            "the neighbouring kernels on replay on this ROCm stack; they come from torch ops that call hipMemsetAsync (ATen's "
            "multi-block reductions: x.max(dim) / x.sum(dim) over a long dimension).  Replace that op by a library kernel "
            "(ops.colmax_arg, ops.colsum_objects, ...) or move it out of the captured region." % (what, memsets, kernels))
    graph.instantiate()
    return kernels, memcpys, memsets, other


class PinnedRing(object):
    """Host staging for the per-forward subsample indices.  A replay is enqueued asynchronously, so the host may be several
    steps ahead of the device: each upload takes the next of `slots` pinned buffers and first waits for the copy that last
    read that buffer (an event), instead of overwriting a buffer an in-flight copy may still be reading."""

    def __init__(self, numel, slots=4):
        self.bufs = [torch.empty(numel, dtype=torch.int32).pin_memory() for _ in range(slots)]
        self.done = [None] * slots
        self.at = 0

    def upload(self, values, dst):
        i = self.at
        self.at = (i + 1) % len(self.bufs)
        if self.done[i] is not None:
            self.done[i].synchronize()
        self.bufs[i].copy_(values.to(torch.int32))
        dst.copy_(self.bufs[i], non_blocking=True)
        self.done[i] = torch.cuda.Event()
        self.done[i].record(torch.cuda.current_stream(dst.device))


class GraphedForward(object):
    def __init__(self, pk, B, N, device, train_keys=False, parts=1, kmax=20, n_cls=6, outputs_only=None):
        global SIDE_TAG
        if parts not in (1, 2) or (parts == 2 and B % 2):
            raise ValueError("parts must be 1, or 2 with an even batch")
        self.B, self.N, self.device = B, N, device
        n1 = N // 4
        n2 = n1 // 4
        self.points = torch.zeros(B, N, 3, device=device)
        self.obj = torch.zeros(B, 1, device=device)
        self.s1 = torch.zeros(n1, device=device, dtype=torch.int32)
        self.s2 = torch.zeros(n2, device=device, dtype=torch.int32)
        self._pins = PinnedRing(n1 + n2)
        self._s12 = torch.zeros(n1 + n2, device=device, dtype=torch.int32)
        half = B // parts

        def run():
            self.s1.copy_(self._s12[:n1])
            self.s2.copy_(self._s12[n1:])
            if parts == 1:
                return posenet_forward(pk, self.points, self.obj, train_keys, (self.s1, self.s2), None, None, kmax, n_cls,
                                       outputs_only=outputs_only)
            global SIDE_TAG, BRANCH_STREAMS
            branch, BRANCH_STREAMS = BRANCH_STREAMS, False     # two streams in all: the halves overlap each other
            cur = torch.cuda.current_stream(device)
            other = _side_stream(device, "half")
            fork = torch.cuda.Event()
            fork.record(cur)
            outs = []
            for part, stream in ((0, cur), (1, other)):
                SIDE_TAG = part
                with torch.cuda.stream(stream):
                    if part:
                        stream.wait_event(fork)
                    sl = slice(part * half, (part + 1) * half)
                    outs.append(posenet_forward(pk, self.points[sl], self.obj[sl], train_keys, (self.s1, self.s2), None, None,
                                                kmax, n_cls, outputs_only=outputs_only))
                    if part:
                        join = torch.cuda.Event()
                        join.record(stream)
            SIDE_TAG, BRANCH_STREAMS = 0, branch
            cur.wait_event(join)
            return {k: torch.cat([outs[0][k], outs[1][k]], 0) for k in outs[0]}

        warm = torch.cuda.Stream(device=device)
        warm.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(warm):           # first-use work (LDS limits, lazy module loads) must not happen under capture
            run()
            run()
        torch.cuda.current_stream(device).wait_stream(warm)
        torch.cuda.synchronize(device)
        self.graph = new_graph()
        with torch.cuda.graph(self.graph):
            self.out = run()
        self.nodes = check_capture(self.graph, "GraphedForward")

    def __call__(self, points, obj_id, sample_idx=None):
        if tuple(points.shape) != (self.B, self.N, 3):
            raise ValueError("GraphedForward was captured for points of shape (%d, %d, 3)" % (self.B, self.N))
        if sample_idx is None:
            sample_idx = draw_sample_idx(self.N)
        self._pins.upload(torch.cat([sample_idx[0].reshape(-1), sample_idx[1].reshape(-1)]), self._s12)
        self.points.copy_(points, non_blocking=True)
        self.obj.copy_(obj_id.reshape(self.B, 1).to(self.points.dtype), non_blocking=True)
        self.graph.replay()
        return self.out
