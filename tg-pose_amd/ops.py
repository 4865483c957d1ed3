"""Tensor-level wrappers over the C ABI: pointer/stride plumbing only, no compute.

Every function enqueues HIP kernels on torch's current stream of the tensors' device and returns
device tensors; nothing synchronises.  Inputs must be fp32 (indices int32), on a GPU, with the
layouts documented in include/tgpose.h -- violations raise instead of being silently copied.
"""
import ctypes
import functools
import os
import math

import torch

from . import _lib
from ._lib import GemmArgs, check


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# bench.py sets this to a dict to time, with HIP events on the launch stream, every launch of the graph / HBM-bound kernel
# class (kNN, graph convolution, ORL pooling, pooling, gathers, row sort, tails, per-object post-processing): class name ->
# list of (start event, end event).  None (the default) adds nothing to a call.
CLASS_TIMER = None


def _timed(cls):
    def deco(fn):
        @functools.wraps(fn)
        def wrap(*a, **kw):
            if CLASS_TIMER is None:
                return fn(*a, **kw)
            st = torch.cuda.current_stream(next(t for t in a if torch.is_tensor(t)).device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            r = fn(*a, **kw)
            e1.record(st)
            CLASS_TIMER.setdefault(cls, []).append((e0, e1, fn.__name__))
            return r
        return wrap
    return deco


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t, name, dims=None):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError("%s must be a float32 GPU tensor" % name)
    if dims is not None and t.dim() != dims:
        raise ValueError("%s must have %d dims" % (name, dims))
    return t


def _rows(t, name):
    """(…, C) tensor whose rows are contiguous; returns (tensor, row stride in elements)."""
    _f32(t, name)
    if t.stride(-1) != 1:
        raise ValueError("%s: last dim must be contiguous" % name)
    ld = t.stride(-2)
    if t.dim() == 3 and t.stride(0) != t.shape[1] * ld:
        raise ValueError("%s: batch stride must equal n * row stride" % name)
    return t, ld


def _i32(t, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous int32 GPU tensor" % name)
    return t


def graph_node_counts(raw_graph):
    """(kernel, memcpy, memset, other) node counts of a captured hipGraph; raw_graph = torch.cuda.CUDAGraph(keep_graph=True)
    .raw_cuda_graph() after the capture (host-only census, tgp_graph_node_counts)"""
    counts = (ctypes.c_int * 4)()
    check(_lib.lib().tgp_graph_node_counts(ctypes.c_void_p(int(raw_graph)), counts), "tgp_graph_node_counts")
    return tuple(int(c) for c in counts)


@_timed("graph")
def center(points, zero=None):
    """points (B,n,3) -> (xyz_c (B,n,3), mean (B,3)); zero (a contiguous 4-byte tensor): cleared by the same launch"""
    _f32(points, "points", 3)
    points = points.contiguous()
    B, n, _ = points.shape
    xyz = torch.empty_like(points)
    mean = torch.empty(B, 3, device=points.device, dtype=torch.float32)
    if zero is not None:
        if not zero.is_contiguous() or zero.element_size() != 4 or zero.device != points.device:
            raise ValueError("center: zero must be a contiguous 4-byte tensor on the points' device")
        check(_lib.lib().tgp_center_zero(_p(points), B, n, _p(xyz), _p(mean), _p(zero), zero.numel(), _stream(points)), "tgp_center_zero")
        return xyz, mean
    check(_lib.lib().tgp_center(_p(points), B, n, _p(xyz), _p(mean), _stream(points)), "tgp_center")
    return xyz, mean


@_timed("graph")
def knn_xyz(xyz, k):
    """xyz (B,n,3) contiguous -> idx (B,n,k) int32"""
    _f32(xyz, "xyz", 3)
    if not xyz.is_contiguous() or xyz.shape[2] != 3:
        raise ValueError("xyz must be contiguous (B,n,3)")
    B, n, _ = xyz.shape
    idx = torch.empty(B, n, k, device=xyz.device, dtype=torch.int32)
    check(_lib.lib().tgp_knn_xyz(_p(xyz), B, n, int(k), _p(idx), _stream(xyz)), "tgp_knn_xyz")
    return idx


KNN_FEAT_FORM = int(os.environ.get("TGP_KNN_FEAT_FORM", "0"))     # 0: the library's choice (tgp_knn_feat_form)


@_timed("graph")
def knn_feat(feat, k, workspace=None, form=None, xyz=None):
    """feat (B,n,d) rows contiguous (row stride may exceed d) -> idx (B,n,k) int32.
    xyz (B,n,3): returns (idx, dirs) -- dirs (B,n,k,4) the unit directions to the selected neighbours (tgp_knn_feat_dirs: written by
    the selecting waves; gconv_hs(dirs=...) takes them) or None where the fused kernel does not serve the shape."""
    feat, ld = _rows(feat, "feat")
    B, n, d = feat.shape
    need = _lib.lib().tgp_knn_feat_workspace_bytes(B, n, d)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 3) // 4, device=feat.device, dtype=torch.float32)
    idx = torch.empty(B, n, k, device=feat.device, dtype=torch.int32)
    if xyz is not None:
        _f32(xyz, "xyz", 3)
        dirs = torch.empty(B, n, k, 4, device=feat.device, dtype=torch.float32)
        rc = _lib.lib().tgp_knn_feat_dirs(_p(feat), ld, B, n, d, int(k), _p(idx), _p(workspace), workspace.numel() * workspace.element_size(),
                                          _p(xyz.contiguous()), _p(dirs), _stream(feat))
        if rc == 0:
            return idx, dirs
        if rc != -2:
            check(rc, "tgp_knn_feat_dirs")
        dirs = None
    check(_lib.lib().tgp_knn_feat_form(_p(feat), ld, B, n, d, int(k), _p(idx), _p(workspace),
                                       workspace.numel() * workspace.element_size(), KNN_FEAT_FORM if form is None else int(form),
                                       _stream(feat)), "tgp_knn_feat")
    return idx if xyz is None else (idx, None)


@_timed("graph")
def nn1(target, source):
    """(B,n,3),(B,m,3) -> idx (B,n) int32"""
    _f32(target, "target", 3), _f32(source, "source", 3)
    target, source = target.contiguous(), source.contiguous()
    B, n, _ = target.shape
    m = source.shape[1]
    idx = torch.empty(B, n, device=target.device, dtype=torch.int32)
    check(_lib.lib().tgp_nn1(_p(target), _p(source), B, n, m, _p(idx), _stream(target)), "tgp_nn1")
    return idx


def nn1_pair(target, source1, source2, tail=None):
    """nn1(target, source1), nn1(target, source2) in one launch; tail = (obj_id (B,) float, feat (B,n,ld), col0, n_cls): the launch
    also writes fill_tail(obj_id, target, feat, col0, n_cls)"""
    for t, nm in ((target, "target"), (source1, "source1"), (source2, "source2")):
        _f32(t, nm, 3)
    target, source1, source2 = target.contiguous(), source1.contiguous(), source2.contiguous()
    B, n, _ = target.shape
    i1 = torch.empty(B, n, device=target.device, dtype=torch.int32)
    i2 = torch.empty(B, n, device=target.device, dtype=torch.int32)
    obj_id, feat, col0, n_cls, ld = None, None, 0, 0, 0
    if tail is not None:
        obj_id, feat, col0, n_cls = tail
        feat, ld = _rows(feat, "feat")
        if tuple(feat.shape[:2]) != (B, n) or obj_id.numel() != B or not obj_id.is_contiguous() or obj_id.dtype != torch.float32:
            raise ValueError("nn1_pair: tail = (obj_id (B,) float32, feat (B,n,ld), col0, n_cls)")
    check(_lib.lib().tgp_nn1_pair_tail(_p(target), _p(source1), _p(source2), B, n, source1.shape[1], source2.shape[1], _p(i1), _p(i2),
                                       _p(obj_id), int(n_cls), _p(feat), ld, int(col0), _stream(target)), "tgp_nn1_pair_tail")
    return i1, i2


def normalize_dirs(directions):
    _f32(directions, "directions", 2)
    directions = directions.contiguous()
    out = torch.empty_like(directions)
    check(_lib.lib().tgp_normalize_dirs(_p(directions), directions.shape[1], _p(out), _stream(directions)),
          "tgp_normalize_dirs")
    return out


def normalize_dirs_bwd(directions, grad):
    directions, grad = directions.contiguous(), grad.contiguous()
    out = torch.empty_like(directions)
    check(_lib.lib().tgp_normalize_dirs_bwd(_p(directions), _p(grad), directions.shape[1], _p(out), _stream(directions)),
          "tgp_normalize_dirs_bwd")
    return out


@_timed("graph")
def gconv_surface(xyz, idx, sdn, S, C, out=None, xyz_pad=False):
    """xyz_pad: `out` is a (B, n, C) view of a buffer whose rows are at least C + 4 floats long; columns C..C+3 of every row
    receive the point (x, y, z, 0) -- the STE convolution's operand for the layer's last GEMM"""
    _f32(xyz, "xyz", 3), _i32(idx, "idx")
    B, n, k = idx.shape
    if out is None:
        out = torch.empty(B, n, C, device=xyz.device, dtype=torch.float32)
    out, ldo = _rows(out, "out")
    check(_lib.lib().tgp_gconv_surface_fwd(_p(xyz), _p(idx), _p(sdn), B, n, k, S, C, _p(out), ldo, 1 if xyz_pad else 0,
                                           _stream(xyz)), "tgp_gconv_surface_fwd")
    return out


@_timed("graph")
def gconv_hs(xyz, idx, proj, sdn, S, C, out=None, dirs=None):
    """dirs (B,n,k,4): the unit neighbour directions knn_feat(xyz=...) left beside idx (no direction launch)"""
    _f32(xyz, "xyz", 3), _i32(idx, "idx")
    proj, ldp = _rows(proj, "proj")
    B, n, k = idx.shape
    if out is None:
        out = torch.empty(B, n, C, device=xyz.device, dtype=torch.float32)
    out, ldo = _rows(out, "out")
    if dirs is not None:
        if tuple(dirs.shape) != (B, n, k, 4) or not dirs.is_contiguous():
            raise ValueError("gconv_hs: dirs (B,n,k,4) contiguous expected")
        rc = _lib.lib().tgp_gconv_hs_fwd_dirs(_p(xyz), _p(idx), _p(proj), ldp, _p(sdn), B, n, k, S, C, _p(out), ldo, _p(dirs), _stream(xyz))
        if rc == 0:
            return out
        if rc != -2:
            check(rc, "tgp_gconv_hs_fwd_dirs")
    # scratch for the unit neighbour directions: only the LDS-staged kernel (pooled levels) uses it
    dirs = torch.empty(B * n * k * 4, device=xyz.device, dtype=torch.float32) if n * 7 * 8 * 4 <= 72 * 1024 else None
    check(_lib.lib().tgp_gconv_hs_fwd(_p(xyz), _p(idx), _p(proj), ldp, _p(sdn), B, n, k, S, C, _p(out), ldo,
                                      _p(dirs), _stream(xyz)), "tgp_gconv_hs_fwd")
    return out


@_timed("graph")
def orl_global(feat, idx):
    """feat (B,n,C), idx (B,n,k) -> (B,C)"""
    feat, ldf = _rows(feat, "feat")
    _i32(idx, "idx")
    B, n, C = feat.shape
    k = idx.shape[2]
    partial = torch.empty(_lib.lib().tgp_orl_partial_floats(B, n, C), device=feat.device, dtype=torch.float32)
    out = torch.empty(B, C, device=feat.device, dtype=torch.float32)
    check(_lib.lib().tgp_orl_global(_p(feat), ldf, _p(idx), B, n, k, C, _p(partial), _p(out), _stream(feat)),
          "tgp_orl_global")
    return out


@_timed("graph")
def orl_rowbias(feat, idx, w2t, planes=None, xyz_tile=None, tickets=None):
    """feat (B,n,C), idx (B,n,k), w2t (C,C) = W2^T -> rb (B,C) = mean_i max_j feat[idx] @ W2^T.
    planes (a Planes of B*n rows): feat is also written as fp16 planes -- it is the A operand of the layer's last GEMM -- by the
    kernel that stages it in LDS anyway (tgp_orl_rowbias_planes; xyz_tile (B,n,3): one more K-tile (x, y, z, 0 ...) behind the C
    channels).  Returns (rb, planes or None): None where that form does not serve the shape and nothing was written.
    tickets (B int32 zeros, handed back as zeros): the one-launch form (tgp_orl_rowbias_fused) where the shape allows."""
    feat, ldf = _rows(feat, "feat")
    _i32(idx, "idx")
    B, n, C = feat.shape
    k = idx.shape[2]
    rb = torch.empty(B, C, device=feat.device, dtype=torch.float32)
    if tickets is not None and ORL_FUSED:
        contrib = torch.empty(B * (C // 16) * C, device=feat.device, dtype=torch.float32)
        rc = _lib.lib().tgp_orl_rowbias_fused(_p(feat), ldf, _p(idx), B, n, k, C, _p(w2t), None, _p(rb),
                                              _p(planes.buf) if planes is not None else None, planes.kt if planes is not None else 0,
                                              _p(planes.amax) if planes is not None else None, _p(xyz_tile), _p(contrib), _p(tickets),
                                              _stream(feat))
        if rc == 0:
            return (rb, planes) if planes is not None else rb
        if rc != -2:
            check(rc, "tgp_orl_rowbias_fused")
    partial = torch.empty(_lib.lib().tgp_orl_partial_floats(B, n, C), device=feat.device, dtype=torch.float32)
    if planes is not None:
        rc = _lib.lib().tgp_orl_rowbias_planes(_p(feat), ldf, _p(idx), B, n, k, C, _p(partial), _p(w2t), None, _p(rb), _p(planes.buf),
                                               planes.kt, _p(planes.amax), _p(xyz_tile), _stream(feat))
        if rc == 0:
            return rb, planes
        if rc != -2:
            check(rc, "tgp_orl_rowbias_planes")
    check(_lib.lib().tgp_orl_rowbias(_p(feat), ldf, _p(idx), B, n, k, C, _p(partial), _p(w2t), None, _p(rb), _stream(feat)),
          "tgp_orl_rowbias")
    return (rb, None) if planes is not None else rb


@_timed("graph")
def pool(xyz, feat, idx, sample, kpool=4, out_f=None, planes=None):
    """xyz (B,n,3), feat (B,n,C), idx (B,n,>=kpool) int32, sample (n_out,) int32 -> (xyz_p, feat_p); planes (a Planes of B*n_out
    rows): feat_p also as fp16 planes, by the same kernel where the shape allows, else by a split of the result"""
    _f32(xyz, "xyz", 3)
    feat, ldf = _rows(feat, "feat")
    _i32(idx, "idx"), _i32(sample, "sample")
    B, n, C = feat.shape
    n_out = sample.numel()
    out_xyz = torch.empty(B, n_out, 3, device=xyz.device, dtype=torch.float32)
    if out_f is None:
        out_f = torch.empty(B, n_out, C, device=xyz.device, dtype=torch.float32)
    out_f, ldo = _rows(out_f, "out_f")
    if planes is not None:
        rc = _lib.lib().tgp_pool_fwd_planes(_p(xyz), _p(feat), ldf, _p(idx), idx.shape[2], _p(sample), B, n, n_out, kpool, C,
                                            _p(out_xyz), _p(out_f), ldo, _p(planes.buf), planes.kt, _p(planes.amax), _stream(xyz))
        if rc == 0:
            return out_xyz, out_f
        if rc != -2:
            check(rc, "tgp_pool_fwd_planes")
    check(_lib.lib().tgp_pool_fwd(_p(xyz), _p(feat), ldf, _p(idx), idx.shape[2], _p(sample), B, n, n_out, kpool, C,
                                  _p(out_xyz), _p(out_f), ldo, _stream(xyz)), "tgp_pool_fwd")
    if planes is not None:
        planes_split.__wrapped__(out_f.view(B * n_out, -1), K=C, out=planes)
    return out_xyz, out_f


@_timed("graph")
def gather_rows(src, idx, dst):
    """dst[b,i,:C] = src[b, idx[b,i], :C]; src (B,n_src,C), idx (B,n_out) int32, dst (B,n_out,C) view"""
    src, lds = _rows(src, "src")
    dst, ldd = _rows(dst, "dst")
    _i32(idx, "idx")
    B, n_src, C = src.shape
    n_out = idx.shape[1]
    check(_lib.lib().tgp_gather_rows(_p(src), lds, _p(idx), B, n_src, n_out, C, _p(dst), ldd, _stream(src)),
          "tgp_gather_rows")
    return dst


@_timed("graph")
def sort_by_parent(near1, near2, n1, n2):
    """Rows of each object sorted by (near2, near1), ties in point order (= torch.argsort(near2 * n1 + near1, stable=True)).
    near1, near2 (B,n) int32 -> order (B,n) int32, order64 (B,n) int64, near1 + b*n1 and near2 + b*n2 in the sorted order."""
    _i32(near1, "near1"), _i32(near2, "near2")
    B, n = near1.shape
    if near2.shape != (B, n):
        raise ValueError("sort_by_parent: near1 and near2 must both be (B,n)")
    order, o1, o2 = torch.empty_like(near1), torch.empty_like(near1), torch.empty_like(near1)
    order64 = torch.empty(B, n, device=near1.device, dtype=torch.int64)
    check(_lib.lib().tgp_sort_by_parent(_p(near1), _p(near2), B, n, n1, n2, _p(order), _p(order64), _p(o1), _p(o2), _stream(near1)),
          "tgp_sort_by_parent")
    return order, order64, o1, o2


@_timed("graph")
def fill_tail(obj_id, xyz_c, feat, col0, n_cls):
    _f32(obj_id, "obj_id"), _f32(xyz_c, "xyz_c", 3)
    feat, ld = _rows(feat, "feat")
    B, n, _ = xyz_c.shape
    check(_lib.lib().tgp_fill_tail(_p(obj_id.contiguous()), _p(xyz_c), B, n, n_cls, _p(feat), ld, col0, _stream(feat)),
          "tgp_fill_tail")


# How launches large enough for the tile kernels run when the caller supplies pre-split weights:
#   "split16"  two-term fp16 operand split, 3 MFMA terms (products to ~2^-22; operands must stay below 65504)
#   "split"    three-term bf16 operand split, 6 MFMA terms (fp32 range, products to ~2^-23)
#   "fp32"     always the fp32 MFMA kernels
# split_w() packs weights for the mode current at pack time; gemm() reads the kind off the packed tensor's shape.
GEMM_MODE = "split16"
PLANES = os.environ.get("TGP_PLANES", "1") != "0"     # activations also as fp16 planes, consumers on the pre-split kernel (split16 mode)
ORL_FUSED = os.environ.get("TGP_ORL_FUSED", "1") != "0"      # the ORL pooling, its mean and its projection as one launch
HEADS_PLANES = os.environ.get("TGP_HEADS_PLANES", "1") != "0"   # measurement switch of the fused kernels: fragments from the fine planes


FP16_SAFE = 32768.0          # |w| at or above this is pre-scaled before the fp16 split (fp16's largest finite value is 65504)


def split_w(W, check=True):
    """Pack-time operand split of a weight for the tile GEMM kernels, in the current GEMM_MODE.
    check (fp16 split only): fp16 holds magnitudes below 65504, so the weight's largest magnitude is read back ONCE here (pack time
    is outside every timed or captured region) and a weight that reaches 2^15 is split as W * 2^-k with the power of two 2^k
    attached to the result (``.tgp_unscale``, a device scalar that ops.gemm hands to the kernel as c_scale: the accumulated sums
    are multiplied back, exactly).  check=False is for splits made INSIDE a captured step (the training path splits the live
    parameters every step): nothing may be read back there, and a parameter beyond fp16's range gives inf / NaN, which the
    trainer's NaN test (trainer/RL_TDA.py:217) sees.  Activations need no such care: the kernels guard them tile by tile."""
    if GEMM_MODE != "split16":
        return split_bf16(W)
    if not check:
        return split_f16(W)
    amax = float(W.detach().abs().max()) if W.numel() else 0.0
    if not (amax >= FP16_SAFE) or not math.isfinite(amax):
        return split_f16(W)
    k = math.frexp(amax)[1] - 15                     # amax * 2^-k lies in [2^14, 2^15)
    out = split_f16(W * (2.0 ** -k))
    out.tgp_unscale = torch.full((1,), 2.0 ** k, device=W.device, dtype=torch.float32)
    return out


def split_rows(ws, a, b=None):
    """rows [a, b) of a pack-time split weight, keeping the power of two a pre-scaled weight carries (split_w's .tgp_unscale is a
    Python attribute that plain slicing would drop: the product would silently come out 2^k too small)"""
    out = ws[a:b]
    unscale = getattr(ws, "tgp_unscale", None)
    if unscale is not None:
        out.tgp_unscale = unscale
    return out


def split_f16(W):
    """W (..., rows, K) fp32 -> int16 tensor (..., rows, ldo // 16, 2, 16) of fp16 bit patterns (hi, lo per K-tile)."""
    W = W.contiguous()
    K = W.shape[-1]
    rows = W.numel() // K
    ldo = (K + 15) // 16 * 16
    out = torch.empty(tuple(W.shape[:-1]) + (ldo // 16, 2, 16), device=W.device, dtype=torch.int16)
    check(_lib.lib().tgp_split_f16(_p(W), rows, K, K, _p(out), ldo, _stream(W)), "tgp_split_f16")
    return out


def split_bf16(W):
    """W (..., rows, K) fp32 -> int16 tensor (..., rows, ldo // 16, 3, 16) of bf16 bit patterns: per 16-wide K-tile the
    hi / mid / lo terms consumed by the split GEMM kernel (ldo = K rounded up to 16, zero padded)."""
    W = W.contiguous()
    K = W.shape[-1]
    rows = W.numel() // K
    ldo = (K + 15) // 16 * 16
    out = torch.empty(tuple(W.shape[:-1]) + (ldo // 16, 3, 16), device=W.device, dtype=torch.int16)
    check(_lib.lib().tgp_split_bf16(_p(W), rows, K, K, _p(out), ldo, _stream(W)), "tgp_split_bf16")
    return out


class Planes(object):
    """A (rows, K) fp32 matrix as BLOCKED fp16 hi / lo planes (include/tgpose.h, tgp_gemm_args.A_planes): what the pre-split GEMM
    kernel (csrc/gemm_pp.hip) stages by LDS-DMA.  buf: uint8 (ceil(rows / 32), kt, 2048); amax: int32 (ceil(rows / 32),) bits of the
    largest magnitude per row block (the fp16 range guard of the consumer), or None for weights (checked once at pack time)."""

    __slots__ = ("buf", "amax", "rows", "K", "kt", "tgp_unscale")

    def __init__(self, rows, K, device, kt=None, amax=True, buf=None, amax_buf=None):
        self.rows, self.K = int(rows), int(K)
        self.kt = int(kt) if kt is not None else (self.K + 15) // 16
        nblk = (self.rows + 31) // 32
        self.buf = buf if buf is not None else torch.empty(nblk, self.kt, 2048, device=device, dtype=torch.uint8)
        self.amax = (amax_buf if amax_buf is not None else torch.zeros(nblk, device=device, dtype=torch.int32)) if amax else None
        self.tgp_unscale = None

    def to_float(self):
        """(tests) the fp32 matrix hi + lo the planes stand for, (rows, K)"""
        nblk = self.buf.shape[0]
        h = self.buf.view(torch.float16).view(nblk, self.kt, 2, 2, 32, 8).float()      # [rb][kt][plane][h][r][8]
        x = (h[:, :, 0] + h[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(nblk * 32, self.kt * 16)
        return x[: self.rows, : self.K]


def planes_on():
    """activations also travel as fp16 planes and their consumers run on the pre-split kernel (the default arithmetic only)"""
    return PLANES and GEMM_MODE == "split16"


@_timed("graph")
def planes_split(X, K=None, kt=None, amax=True, out=None):
    """X (..., K) fp32 rows (row stride >= K) -> Planes (tgp_planes_split).  Weights at pack time (amax=False), activations whose
    producer does not write planes itself."""
    X, ld = _rows(X, "X")
    K = X.shape[-1] if K is None else K
    rows = math.prod(X.shape[:-1])
    P = out if out is not None else Planes(rows, K, X.device, kt=kt, amax=amax)
    check(_lib.lib().tgp_planes_split(_p(X), rows, K, ld, _p(P.buf), P.kt, _p(P.amax), _stream(X)), "tgp_planes_split")
    return P


@_timed("graph")
def planes_gather(src, idx, dst, K, planes):
    """gather_rows(src, idx, dst) that also writes the planes of the gathered rows' first K columns (tgp_planes_gather)"""
    src, lds = _rows(_f32(src, "src", 3), "src")
    dst, ldd = _rows(_f32(dst, "dst", 3), "dst")
    B, n_src, C = src.shape
    n_out = dst.shape[1]
    check(_lib.lib().tgp_planes_gather(_p(src), lds, _p(_i32(idx, "idx")), B, n_src, n_out, K, C, _p(dst), ldd, _p(planes.buf),
                                       planes.kt, _p(planes.amax), _stream(src)), "tgp_planes_gather")
    return dst


def planes_w(W):
    """pack-time planes of a weight (N, K) for the pre-split kernel, with split_w's range rule: a weight that reaches 2^15 is split
    as W * 2^-k and the power of two rides along as .tgp_unscale (c_scale of the launch)."""
    amax = float(W.detach().abs().max()) if W.numel() else 0.0
    if not (amax >= FP16_SAFE) or not math.isfinite(amax):
        return planes_split(W.contiguous(), amax=False)
    k = math.frexp(amax)[1] - 15
    P = planes_split((W * (2.0 ** -k)).contiguous(), amax=False)
    P.tgp_unscale = torch.full((1,), 2.0 ** k, device=W.device, dtype=torch.float32)
    return P


# bench.py sets this to a list to time, with HIP events on the launch stream, every launch that the
# library routes to its 128x128-tile MFMA kernel (same rule as tgp_gemm_f32 in csrc/gemm.hip).
GEMM_TIMER = None
GEMM_TIMER_ALL = False      # development: time the small-tile and skinny launches too (scripts/gemm_shapes.py)


def _routes_to_big_tile(M, N, batch=1, split16=False):
    """does tgp_gemm_f32 send this launch to a tile kernel that reads the SPLIT weight (rather than to the exact-fp32 64 x 64 or the
    skinny kernel, which read the fp32 weight)?  split16: a two-term fp16 split weight is passed -- such launches also take the
    64 x 128 small-tile split kernel from 32 tiles on (round 3, csrc/gemm.hip `small_split`)."""
    if not (M > 32 and N > 64):
        return False
    if ((M + 127) // 128) * ((N + 127) // 128) * batch >= _big_tile_threshold():
        return True
    return bool(split16) and ((M + 63) // 64) * ((N + 127) // 128) * batch >= 32


_BIG_THR = None


def _big_tile_threshold():
    """resident_slots() / 2 of csrc/gemm.hip (slots = CUs: one 1024-thread workgroup per CU)"""
    global _BIG_THR
    if _BIG_THR is None:
        _BIG_THR = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count // 2
    return _BIG_THR


def gemm(A, W, C=None, *, M=None, N=None, K=None, lda=None, ldw=None, ldc=None, bias=None, rowbias=None,
         rows_per_obj=0, res1=None, ldr1=0, res2=None, ldr2=0, scale=None, shift=None, act=0, slope=0.0,
         colmax_keys=None, k_alg=None, slope_vec=None, cm_cols=0, c_col0=0, batch=1, batch_strides=None, w_split=None,
         a_scale=None, c_scale=None, ksplit_chunk=0, gather1=None, gather2=None, flops_ref=None, epilogue=0, pred=None,
         row_base=0, a_planes=None, w_planes=None, c_planes=None, cp_col0=0, pp_config=0, range_flag=None, a_keys=False, a_wrap=0,
         c_sigmoid=None):
    """Raw call into tgp_gemm_f32.  A/W/C/res* are tensors whose data_ptr is the first element of the
    operand (views into wider buffers are fine); all sizes/strides are explicit.  k_alg: the layer's
    true input width when K includes zero padding (only used for FLOP accounting in bench.py)."""
    split16 = w_split is not None and GEMM_MODE != "fp32" and w_split.shape[-2] == 2 and not ksplit_chunk
    # the pre-split kernel serves the launches the split tile kernels serve (so that results do not depend on whether planes were
    # handed in: launches too small for them run on the exact-fp32 small kernel either way)
    tile = _routes_to_big_tile(M, N, batch, True) and not ksplit_chunk
    pp = a_planes is not None and w_planes is not None and GEMM_MODE == "split16" and PLANES and tile and w_split is not None
    late_planes = None
    if c_planes is not None and GEMM_MODE == "split16" and PLANES and not (tile and w_split is not None):
        late_planes, c_planes = c_planes, None          # a small launch: its result is split by a launch of its own below
    timed = GEMM_TIMER is not None and pred is None and (GEMM_TIMER_ALL or pp or _routes_to_big_tile(M, N, batch, split16))   # (a predicated
    # launch is a repair path that normally does nothing: it has no place in a FLOP rate)
    if A is None:                                # a planes-only operand (tgp_gemm_args.range_flag)
        if not pp or range_flag is None:
            raise ValueError("gemm: an operand without its fp32 form needs both operands' planes, a tile-kernel launch and a range flag")
        A = a_planes.buf                         # (device / stream of the launch)
        a_ptr, lda = None, 0
    else:
        a_ptr = _p(A)
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(A.device))
    a = GemmArgs()
    a.A, a.lda, a.W, a.ldw = a_ptr, lda, _p(W), ldw
    a.C, a.ldc = (_p(C), ldc) if C is not None else (None, 0)
    a.M, a.N, a.K = M, N, K
    a.bias = _p(bias)
    a.rowbias, a.ldrb = (_p(rowbias), rowbias.stride(0)) if rowbias is not None else (None, 0)
    a.rows_per_obj = rows_per_obj
    a.res1, a.ldr1 = _p(res1), ldr1
    a.res2, a.ldr2 = _p(res2), ldr2
    a.scale, a.shift = _p(scale), _p(shift)
    a.act, a.slope = act, slope
    a.colmax_keys, a.ldcm = (_p(colmax_keys), colmax_keys.stride(-2)) if colmax_keys is not None else (None, 0)
    a.slope_vec, a.cm_cols, a.c_col0, a.batch = _p(slope_vec), cm_cols, c_col0, batch
    if batch_strides is not None:
        (a.batch_stride_a, a.batch_stride_w, a.batch_stride_c, a.batch_stride_vec, a.batch_stride_colmax) = batch_strides
    if w_split is not None and GEMM_MODE != "fp32":
        a.W_split, a.ldws = _p(w_split), w_split.shape[-3] * 16
        a.w_split_kind = 1 if w_split.shape[-2] == 2 else 0
        unscale = getattr(w_split, "tgp_unscale", None)       # a weight that was pre-scaled into fp16's range (split_w)
        if unscale is not None and not pp and _routes_to_big_tile(M, N, batch, split16):
            c_scale = unscale if c_scale is None else c_scale * unscale
    if pp and w_planes.tgp_unscale is not None:
        c_scale = w_planes.tgp_unscale if c_scale is None else c_scale * w_planes.tgp_unscale
    a.a_scale, a.c_scale, a.ksplit_chunk = _p(a_scale), _p(c_scale), int(ksplit_chunk)
    if gather1 is not None:                      # (rows tensor whose data_ptr is the first column wanted, row stride, int32 row ids)
        a.gres1, a.ldg1, a.gidx1 = _p(gather1[0]), int(gather1[1]), _p(gather1[2])
    if gather2 is not None:
        a.gres2, a.ldg2, a.gidx2 = _p(gather2[0]), int(gather2[1]), _p(gather2[2])
    a.epilogue = int(epilogue)
    a.pred = _p(pred)
    a.row_base = int(row_base)
    if pp:                                       # both operands as blocked fp16 planes: the pre-split kernel (bit-identical results)
        a.A_planes, a.a_kt, a.a_amax = _p(a_planes.buf), a_planes.kt, _p(a_planes.amax)
        a.W_planes, a.w_kt = _p(w_planes.buf), w_planes.kt
        a.pp_config = int(pp_config)
        a.range_flag = _p(range_flag)
    if c_planes is not None and GEMM_MODE == "split16" and PLANES:
        a.C_planes, a.c_kt, a.cp_col0, a.c_amax = _p(c_planes.buf), c_planes.kt, int(cp_col0), _p(c_planes.amax)
    a.a_keys, a.a_wrap, a.C_sigmoid = int(bool(a_keys)), int(a_wrap), _p(c_sigmoid)
    check(_lib.lib().tgp_gemm_f32(ctypes.byref(a), _stream(A)), "tgp_gemm_f32")
    if late_planes is not None:
        if C is None or batch != 1 or (cp_col0 & 15):
            raise ValueError("gemm: result planes of a small launch need its fp32 result")
        Kc = N - c_col0
        if Kc % 16 and (cp_col0 // 16 + (Kc + 15) // 16) < late_planes.kt:
            # the split kernel zero-fills the tail of its last K-tile: inside a wider planes buffer (fm2 | fm3) that tile belongs to
            # the neighbouring column range (the C-side rule of the C_planes path, csrc/gemm.hip)
            raise ValueError("gemm: result planes into a column range of a wider buffer need a multiple of 16 columns")
        check(_lib.lib().tgp_planes_split_cols(_p(C), M, Kc, ldc, _p(late_planes.buf), late_planes.kt, int(cp_col0), _p(late_planes.amax),
                                               _stream(A)), "tgp_planes_split_cols")
    if timed:
        e1.record(torch.cuda.current_stream(A.device))
        # flops_ref: what this launch stands for in the reference's formulation (the factored wide layers run fewer FLOPs)
        fl = 2.0 * M * N * (k_alg or K) * batch
        # algorithmic bytes of the launch (SURVEY 8d's rule: every operand once): A, W, the stored result (and its planes), plain and
        # gathered residuals -- bench.py sets them against the PMC traffic of the tile-GEMM kernel family
        nb = batch * 4.0 * (M * K + N * K + (M * (N - c_col0) if C is not None else 0) + (M * (N - c_col0) if c_planes is not None else 0)
                            + (M * N if res1 is not None else 0) + (M * N if res2 is not None else 0)
                            + (gather1[0].shape[0] * N if gather1 is not None else 0) + (gather2[0].shape[0] * N if gather2 is not None else 0))
        GEMM_TIMER.append((e0, e1, fl, (M, N, K, batch), fl if flops_ref is None else float(flops_ref), nb, "tile"))
    return C


def linear_rows(x, weight, bias=None, scale=None, shift=None, act=0, slope=0.0, out=None, rowbias=None,
                rows_per_obj=0, res1=None, res2=None, colmax_keys=None, want_out=True, k_alg=None, w_split=None,
                a_scale=None, c_scale=None, flops_ref=None, a_planes=None, w_planes=None, c_planes=None, cp_col0=0, pred=None):
    """x (..., K) rows (row stride >= K), weight (N, Kw>=K) -> (..., N).  Convenience over gemm()."""
    x, lda = _rows(x, "x")
    weight, ldw = _rows(weight, "weight")
    K = x.shape[-1]
    M = math.prod(x.shape[:-1])
    N = weight.shape[0]
    ldc = 0
    if want_out:
        if out is None:
            out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        out, ldc = _rows(out, "out")
    r1 = r2 = None
    l1 = l2 = 0
    if res1 is not None:
        r1, l1 = _rows(res1, "res1")
    if res2 is not None:
        r2, l2 = _rows(res2, "res2")
    gemm(x, weight, out if want_out else None, M=M, N=N, K=K, lda=lda, ldw=ldw, ldc=ldc, bias=bias, rowbias=rowbias,
         rows_per_obj=rows_per_obj, res1=r1, ldr1=l1, res2=r2, ldr2=l2, scale=scale, shift=shift, act=act, slope=slope,
         colmax_keys=colmax_keys, k_alg=k_alg, w_split=w_split, a_scale=a_scale, c_scale=c_scale, flops_ref=flops_ref,
         a_planes=a_planes, w_planes=w_planes, c_planes=c_planes, cp_col0=cp_col0, pred=pred)
    return out


def heads_pack_w2(W2, bias1, scale1, shift1):
    """conv2 weights of the heads (heads, 256, 1024) fp32 and the heads' conv1 bias / BatchNorm scale / shift (heads * 1024 each) ->
    the fused heads kernel's operand: per (head, 32-channel block) conv2's fp16 hi / lo planes in MFMA fragment order (K permuted)
    and the block's three vectors (tgp_heads_pack_w2, ABI 7)"""
    W2 = W2.contiguous()
    heads = W2.shape[0]
    if tuple(W2.shape[1:]) != (256, 1024):
        raise ValueError("heads_pack_w2: (heads, 256, 1024) expected")
    vec = [v.contiguous() for v in (bias1, scale1, shift1)]
    if any(v.numel() != heads * 1024 for v in vec):
        raise ValueError("heads_pack_w2: bias1 / scale1 / shift1 of heads * 1024 channels expected")
    out = torch.empty(_lib.lib().tgp_heads_w2_bytes(heads), device=W2.device, dtype=torch.uint8)
    check(_lib.lib().tgp_heads_pack_w2(_p(W2), _p(vec[0]), _p(vec[1]), _p(vec[2]), heads, _p(out), _stream(W2)), "tgp_heads_pack_w2")
    out.tgp_heads = heads
    return out


def heads_repack_vectors(w2p, bias1, scale1, shift1):
    """rewrite, in place, the conv1 bias / BatchNorm scale / shift inside a heads_pack_w2 operand (the fold moved with the running
    statistics; the weights did not): a captured forward that holds the operand's address follows"""
    vec = [v.contiguous() for v in (bias1, scale1, shift1)]
    check(_lib.lib().tgp_heads_pack_w2(None, _p(vec[0]), _p(vec[1]), _p(vec[2]), w2p.tgp_heads, _p(w2p), _stream(w2p)), "tgp_heads_pack_w2")


def heads_planes_w(Wa_heads):
    """conv1 weights of the heads over the fine buffer (heads * 1024, ld >= 268) -> blocked fp16 planes with 17 K-tiles (the fused heads
    kernel's wa_planes)"""
    Wa_heads, ld = _rows(Wa_heads, "Wa_heads")
    return planes_split(Wa_heads, K=min(Wa_heads.shape[-1], 272), kt=17, amax=False)


def heads_fused(fine, K, wa_planes, p1, idx1, p2, idx2, w2p, bias2, scale2, shift2, B, rows_per_obj, k_alg=None,
                keys=None, overflow=None, rows=0, fine_planes=None):
    """conv1 -> BN -> ReLU -> conv2 -> BN -> ReLU -> max over points of the three heads (tgp_heads_fused): keys (heads, B, 256)
    and the device flag (1,) int32 that a wave raises instead of writing keys when it met a magnitude beyond fp16's range.
    fine (M, ldf); wa_planes: heads_planes_w(conv1 weights of the heads); w2p: heads_pack_w2(...) (conv2 weights + conv1's bias /
    scale / shift); p1 / p2: 2-D views whose column 0 is the first head's first channel (row stride = their .stride(0))."""
    fine, ldf = _rows(fine, "fine")
    heads = w2p.tgp_heads
    M = B * rows_per_obj
    if wa_planes.kt != 17 or wa_planes.rows != heads * 1024:
        raise ValueError("heads_fused: wa_planes must hold heads * 1024 rows in 17 K-tiles (ops.heads_planes_w)")
    if keys is None:          # (the eval forward hands in zeroed slices of its one per-forward arena instead)
        keys = torch.zeros(heads, B, 256, device=fine.device, dtype=torch.int32)
    if overflow is None:
        overflow = torch.zeros(1, device=fine.device, dtype=torch.int32)
    timed = GEMM_TIMER is not None
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(fine.device))
    a = _lib.HeadsFusedArgs()
    a.fine, a.ldf, a.K = _p(fine), ldf, K
    a.wa_planes = _p(wa_planes.buf)
    a.p1, a.ldp1, a.idx1 = _p(p1), p1.stride(0), _p(idx1)
    a.p2, a.ldp2, a.idx2 = _p(p2), p2.stride(0), _p(idx2)
    a.w2p = _p(w2p)
    a.bias2, a.scale2, a.shift2 = _p(bias2), _p(scale2), _p(shift2)
    a.keys = _p(keys)
    a.M, a.rows_per_obj, a.B, a.heads = M, rows_per_obj, B, heads
    a.overflow = _p(overflow)
    a.rows = int(rows)
    if fine_planes is not None and planes_on() and HEADS_PLANES:      # the points' features as fp16 planes: operand fragments load as they lie
        a.fine_planes, a.fine_kt, a.fine_amax = _p(fine_planes.buf), fine_planes.kt, _p(fine_planes.amax)
    check(_lib.lib().tgp_heads_fused(ctypes.byref(a), _stream(fine)), "tgp_heads_fused")
    if timed:
        e1.record(torch.cuda.current_stream(fine.device))
        Mk = rows if rows else M                                     # rows this launch processed
        conv2 = 2.0 * Mk * 256 * 1024 * heads
        # algorithmic bytes (SURVEY 8d's rule, every operand once): the points' features, the heads' columns of both coarse products,
        # the weights (conv1 planes + the packed conv2 image), the keys
        nb = 4.0 * (Mk * 272 + (p1.shape[0] + p2.shape[0]) * heads * 1024 + heads * 1024 * 272 + heads * 256 * 1024 + heads * B * 256)
        GEMM_TIMER.append((e0, e1, 2.0 * Mk * heads * 1024 * K + conv2, (Mk, heads * 1024, K, 1),
                           2.0 * Mk * heads * 1024 * (k_alg or K) + conv2, nb, "fused"))
    return keys, overflow


def dec_pack(w2, w3, w4, h1_permuted=False):
    """the decoder's 512 -> 512, 512 -> 256, 256 -> 128 weights -> the fused decoder kernel's staging image (tgp_dec_pack).
    h1_permuted: the operand will come from dec_l1 (channels in accumulator order) instead of a tile GEMM's C_planes"""
    if tuple(w2.shape) != (512, 512) or tuple(w3.shape) != (256, 512) or tuple(w4.shape) != (128, 256):
        raise ValueError("dec_pack: weights of (512, 512), (256, 512), (128, 256) expected")
    out = torch.empty(_lib.lib().tgp_dec_pack_bytes(), device=w2.device, dtype=torch.uint8)
    check(_lib.lib().tgp_dec_pack(_p(w2.contiguous()), _p(w3.contiguous()), _p(w4.contiguous()), int(bool(h1_permuted)), _p(out),
                                  _stream(w2)), "tgp_dec_pack")
    out.tgp_h1_permuted = bool(h1_permuted)
    return out


def proj_pack(w):
    """W (N, K) of a projection GEMM -> the staging image of tgp_proj_planes; None for shapes the kernel does not serve
    (K = 128 / 256 / 512, N % 128 == 0)"""
    N, K = w.shape
    nbytes = _lib.lib().tgp_proj_pack_bytes(int(K), int(N))
    if nbytes < 0:
        return None
    w = w.contiguous()
    out = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    check(_lib.lib().tgp_proj_pack(_p(w), K, K, N, _p(out), _stream(w)), "tgp_proj_pack")
    out.tgp_shape = (int(K), int(N))
    return out


def proj_planes(a_planes, units, bias, x, weight, out=None, flops_ref=None):
    """x W^T (+ bias) on the projection kernel (tgp_proj_planes): a_planes = the planes of x (its first K columns), units =
    proj_pack(weight); x (..., K) fp32 rows and weight (N, K) are read only by tiles the fp16 range rule sends to the exact path."""
    K, N = units.tgp_shape
    x, lda = _rows(x, "x")
    weight, ldw = _rows(weight, "weight")
    M = a_planes.rows
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    o, ldc = _rows(out, "out")
    timed = GEMM_TIMER is not None
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(x.device))
    a = _lib.ProjPlanesArgs()
    a.a_planes, a.a_kt, a.a_amax = _p(a_planes.buf), a_planes.kt, _p(a_planes.amax)
    a.a, a.lda, a.M, a.K, a.N = _p(x), lda, M, K, N
    a.units, a.w, a.ldw, a.bias = _p(units), _p(weight), ldw, _p(bias)
    a.c, a.ldc = _p(o), ldc
    check(_lib.lib().tgp_proj_planes(ctypes.byref(a), _stream(x)), "tgp_proj_planes")
    if timed:
        e1.record(torch.cuda.current_stream(x.device))
        fl = 2.0 * M * N * K
        GEMM_TIMER.append((e0, e1, fl, (M, N, K, 1), fl if flops_ref is None else float(flops_ref), 4.0 * (M * K + N * K + M * N), "fused"))
    return out


def hs_chain_pack(w1, w2):
    """W1 (N1, K1) of an HS layer's last GEMM and W2 (N2, N1) of the next layer's projection -> the fused pair's staging image
    (tgp_hs_chain_pack); None for shapes the kernel does not serve"""
    (N1, K1), (N2, k2) = w1.shape, w2.shape
    nbytes = _lib.lib().tgp_hs_chain_pack_bytes(int(K1), int(N1), int(N2)) if k2 == N1 else -1
    if nbytes < 0:
        return None
    w1, w2 = w1.contiguous(), w2.contiguous()
    out = torch.empty(nbytes, device=w1.device, dtype=torch.uint8)
    check(_lib.lib().tgp_hs_chain_pack(_p(w1), K1, K1, N1, _p(w2), N1, N2, _p(out), _stream(w1)), "tgp_hs_chain_pack")
    out.tgp_shape = (int(K1), int(N1), int(N2))
    return out


def hs_chain(a_planes, units, c1, bias2, flag, rowbias=None, rows_per_obj=0, res1=None, res2=None, scale1=None, shift1=None, relu=True,
             c1_planes=None, c1_col0=0, c2=None):
    """c1 = act(bn(A W1^T + rowbias[object] + res1 + res2)), c2 = c1 W2^T + bias2 in ONE launch (tgp_hs_chain): an HS layer's last
    GEMM and the next layer's projection.  a_planes: Planes of A (M rows); units: hs_chain_pack(W1, W2); c1 (M, N1) view (row stride
    = its .stride(-2)), written; c1_planes: Planes that also receive c1 from K-column c1_col0; c2 (M, N2) (allocated when None); flag:
    (1,) int32 zeroed by the caller, raised when the fp16 split's range is left -- the caller's two tile-kernel launches must follow,
    predicated on it.  Returns c2."""
    K1, N1, N2 = units.tgp_shape
    M = a_planes.rows
    c1r, ldc1 = _rows(c1, "c1")
    if c2 is None:
        c2 = torch.empty(M, N2, device=c1.device, dtype=torch.float32)
    timed = GEMM_TIMER is not None
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(c1.device))
    a = _lib.HsChainArgs()
    a.a_planes, a.a_kt, a.a_amax = _p(a_planes.buf), a_planes.kt, _p(a_planes.amax)
    a.M, a.K1, a.N1, a.N2 = M, K1, N1, N2
    a.units = _p(units)
    if rowbias is not None:
        a.rowbias, a.ldrb, a.rows_per_obj = _p(rowbias), rowbias.stride(0), rows_per_obj
    if res1 is not None:
        res1, a.ldr1 = _rows(res1, "res1")
        a.res1 = _p(res1)
    if res2 is not None:
        res2, a.ldr2 = _rows(res2, "res2")
        a.res2 = _p(res2)
    a.scale1, a.shift1, a.relu = _p(scale1), _p(shift1), int(bool(relu))
    a.c1, a.ldc1 = _p(c1r), ldc1
    if c1_planes is not None:
        if c1_col0 % 16:
            raise ValueError("hs_chain: c1_col0 must be a multiple of 16")
        a.c1_planes, a.c1_kt, a.c1_kt0, a.c1_amax = _p(c1_planes.buf), c1_planes.kt, c1_col0 // 16, _p(c1_planes.amax)
    a.bias2, a.c2, a.ldc2, a.flag = _p(bias2), _p(c2), c2.stride(-2), _p(flag)
    check(_lib.lib().tgp_hs_chain(ctypes.byref(a), _stream(c1)), "tgp_hs_chain")
    if timed:
        e1.record(torch.cuda.current_stream(c1.device))
        nb = 4.0 * (M * 16 * a_planes.kt + M * N1 * (2 + (res1 is not None) + (res2 is not None)) + M * N2 + N1 * K1 + N2 * N1)
        fl = 2.0 * M * (N1 * K1 + N2 * N1)
        GEMM_TIMER.append((e0, e1, fl, (M, N1 + N2, K1, 1), fl, nb, "fused"))
    return c2


def dec_l1(fine_planes, wa_planes, p1, idx1, p2, idx2, bias, scale, shift, rowbias, rows_per_obj, h1_planes, flag, k_alg=None):
    """The decoder's first conv on the factored form (tgp_dec_l1): fine_planes (M rows) x wa_planes (512 x 272 as planes, 17 K-tiles:
    heads_planes_w) + p1[idx1] + p2[idx2] + rowbias[object], BatchNorm fold, ReLU -> h1_planes (M, 512) in ACCUMULATOR channel order
    (for dec_fused with dec_pack(h1_permuted=True) only).  p1 / p2: 2-D views whose column 0 is the conv's first channel."""
    M = fine_planes.rows
    if wa_planes.kt != 17 or wa_planes.rows != 512 or h1_planes.rows != M or h1_planes.K != 512:
        raise ValueError("dec_l1: wa_planes (512 rows, 17 K-tiles) and h1_planes (M, 512) expected")
    a = _lib.DecL1Args()
    a.fine_planes, a.fine_kt, a.fine_amax = _p(fine_planes.buf), fine_planes.kt, _p(fine_planes.amax)
    a.wa_planes = _p(wa_planes.buf)
    a.p1, a.ldp1, a.idx1 = _p(p1), p1.stride(0), _p(idx1)
    a.p2, a.ldp2, a.idx2 = _p(p2), p2.stride(0), _p(idx2)
    a.bias, a.scale, a.shift = _p(bias), _p(scale), _p(shift)
    if rowbias is not None:
        a.rowbias, a.ldrb = _p(rowbias), rowbias.stride(0)
    a.rows_per_obj = int(rows_per_obj)
    a.h1_planes, a.h1_kt, a.h1_amax = _p(h1_planes.buf), h1_planes.kt, _p(h1_planes.amax)
    a.flag, a.M = _p(flag), M
    timed = GEMM_TIMER is not None
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(p1.device))
    check(_lib.lib().tgp_dec_l1(ctypes.byref(a), _stream(p1)), "tgp_dec_l1")
    if timed:
        e1.record(torch.cuda.current_stream(p1.device))
        nb = 4.0 * (M * 272 + (p1.shape[0] + p2.shape[0]) * 512 + 512 * 272 + M * 512)
        GEMM_TIMER.append((e0, e1, 2.0 * M * 512 * 268, (M, 512, 268, 1), 2.0 * M * 512 * (k_alg or 268), nb, "fused"))
    return h1_planes


def dec_fused(h1_planes, units, vecs, w5, b5, order, rows_per_obj, flag, out=None):
    """The decoder behind its first conv as one launch (tgp_dec_fused): h1_planes = the first conv's activation (M, 512) as planes;
    vecs = ((bias, scale, shift) of the 512 -> 512 layer, of 512 -> 256, of 256 -> 128); w5 (3, 128), b5 (3); order (B, n) int64 (the
    sort's permutation) or None; flag (1,) int32 (zeroed by the caller).  -> (M, 3) rows in point order"""
    M = h1_planes.rows
    if h1_planes.K != 512:
        raise ValueError("dec_fused: a (M, 512) operand expected")
    if out is None:
        out = torch.empty(M, 3, device=units.device, dtype=torch.float32)
    a = _lib.DecFusedArgs()
    a.h1_planes, a.h1_kt, a.h1_amax = _p(h1_planes.buf), h1_planes.kt, _p(h1_planes.amax)
    a.units = _p(units)
    keep = []
    for l in range(3):
        for v in range(3):
            t = vecs[l][v].contiguous()
            keep.append(t)
            a.vec[l][v] = t.data_ptr()
    w5c, b5c = w5.contiguous(), b5.contiguous()
    a.w5, a.b5 = _p(w5c), _p(b5c)
    a.map, a.rows_per_obj = _p(order), int(rows_per_obj)
    a.out, a.flag, a.M = _p(out), _p(flag), M
    timed = GEMM_TIMER is not None
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(units.device))
    check(_lib.lib().tgp_dec_fused(ctypes.byref(a), _stream(units)), "tgp_dec_fused")
    if timed:
        e1.record(torch.cuda.current_stream(units.device))
        fl = 2.0 * M * (512 * 512 + 256 * 512 + 128 * 256 + 3 * 128)
        nb = 4.0 * (M * 512 + 512 * 512 + 256 * 512 + 128 * 256 + M * 3) + 8.0 * M
        GEMM_TIMER.append((e0, e1, fl, (M, 512 + 256 + 128 + 3, 512, 1), fl, nb, "fused"))
    return out


def conv_max_fused(fine, K, wa_planes, p1, idx1, p2, idx2, bias, scale, shift, slope, B, rows_per_obj, k_alg=None, keys=None, overflow=None,
                   fine_planes=None):
    """conv -> BN -> LeakyReLU -> max over points of a factored layer (tgp_conv_max_fused): keys (B, C) and the overflow flag.
    wa_planes: heads_planes_w(the layer's (C, ld >= 268) weight over the fine buffer)."""
    fine, ldf = _rows(fine, "fine")
    C = bias.numel()
    M = B * rows_per_obj
    if keys is None:
        keys = torch.zeros(B, C, device=fine.device, dtype=torch.int32)
    if overflow is None:
        overflow = torch.zeros(1, device=fine.device, dtype=torch.int32)
    timed = GEMM_TIMER is not None
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(fine.device))
    a = _lib.ConvMaxFusedArgs()
    a.fine, a.ldf, a.K = _p(fine), ldf, K
    if wa_planes.kt != 17 or wa_planes.rows != C:
        raise ValueError("conv_max_fused: wa_planes must hold the layer's C rows in 17 K-tiles (ops.heads_planes_w)")
    a.wa_planes = _p(wa_planes.buf)
    a.p1, a.ldp1, a.p1_rows, a.idx1 = _p(p1), p1.stride(0), p1.shape[0], _p(idx1)
    a.p2, a.ldp2, a.p2_rows, a.idx2 = _p(p2), p2.stride(0), p2.shape[0], _p(idx2)
    a.bias, a.scale, a.shift, a.slope = _p(bias), _p(scale), _p(shift), float(slope)
    a.keys, a.ldk = _p(keys), C
    a.M, a.rows_per_obj, a.C = M, rows_per_obj, C
    a.overflow = _p(overflow)
    if fine_planes is not None and planes_on() and HEADS_PLANES:
        a.fine_planes, a.fine_kt, a.fine_amax = _p(fine_planes.buf), fine_planes.kt, _p(fine_planes.amax)
    check(_lib.lib().tgp_conv_max_fused(ctypes.byref(a), _stream(fine)), "tgp_conv_max_fused")
    if timed:
        e1.record(torch.cuda.current_stream(fine.device))
        nb = 4.0 * (M * 272 + (p1.shape[0] + p2.shape[0]) * C + C * 272 + B * C)
        GEMM_TIMER.append((e0, e1, 2.0 * M * C * K, (M, C, K, 1), 2.0 * M * C * (k_alg or K), nb, "fused"))
    return keys, overflow


@_timed("graph")
def colmax_decode(keys, out2=False):
    rows, N = keys.shape
    out = torch.empty(rows, 2 * N if out2 else N, device=keys.device, dtype=torch.float32)
    second = out[:, N:] if out2 else None
    check(_lib.lib().tgp_colmax_decode(_p(keys), keys.stride(0), rows, N, _p(out), out.stride(0), _p(second),
                                       _stream(keys)), "tgp_colmax_decode")
    return out


@_timed("graph")
def colmax(x):
    """x (B,n,C) rows -> (B,C) max over n"""
    x, ld = _rows(x, "x")
    B, n, C = x.shape
    out = torch.empty(B, C, device=x.device, dtype=torch.float32)
    check(_lib.lib().tgp_colmax(_p(x), ld, B, n, C, _p(out), _stream(x)), "tgp_colmax")
    return out


@_timed("graph")
def sigmoid(x):
    _f32(x, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().tgp_sigmoid(_p(x), _p(y), x.numel(), _stream(x)), "tgp_sigmoid")
    return y


@_timed("graph")
def head_post(green, red, ts, mean):
    """green / red (B, >=4), ts (B, >=6): 2-D views with contiguous rows (their row strides are passed on)"""
    B = green.shape[0]
    dev = green.device
    for t in (green, red, ts):
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError("head_post: 2-D inputs with contiguous rows expected")
    pg, pr = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    fg, fr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    pT, ps = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    check(_lib.lib().tgp_head_post(_p(green), _p(red), _p(ts), green.stride(0), red.stride(0), ts.stride(0), _p(mean), B, _p(pg),
                                   _p(pr), _p(fg), _p(fr), _p(pT), _p(ps), _stream(green)), "tgp_head_post")
    return pg, pr, fg, fr, pT, ps


def rows_out(x, w, bias=None, order=None, out=None, pred=None):
    """x (B,n,K), w (n_out <= 4, K), order (B,n) int64 or None -> out (B,n,n_out) with out[b, order[b,i]] = x[b,i] @ w^T + bias
    (tgp_rows_out: a narrow last layer and the scatter that undoes a row sort, one launch).  pred: device int -- the launch returns at
    once while it is 0 (the last step of a repair chain, writing into `out`)"""
    x, ld = _rows(x, "x")
    B, n, K = x.shape
    if not w.is_contiguous() or w.shape[1] != K:
        raise ValueError("rows_out: w (n_out, K) contiguous expected")
    if order is not None and (order.dtype != torch.int64 or not order.is_contiguous() or tuple(order.shape) != (B, n)):
        raise ValueError("rows_out: order (B,n) int64 contiguous expected")
    if out is None:
        out = torch.empty(B, n, w.shape[0], device=x.device, dtype=torch.float32)
    check(_lib.lib().tgp_rows_out_pred(_p(x), ld, B * n, K, _p(w), w.stride(0), _p(bias), w.shape[0], _p(order), n, _p(out), _p(pred),
                                       _stream(x)), "tgp_rows_out")
    return out


def pose_tail(keys2, w3t, b3, scale3, shift3, w4, b4, mean, raw=None):
    """keys2 (3,B,256) int32 max keys of the heads' conv2 -> the six pose outputs (tgp_pose_tail: conv3, conv4 and head_post as one
    launch); raw (3,B,8): conv4's outputs"""
    B, dev = keys2.shape[1], keys2.device
    pg, pr = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    fg, fr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    pT, ps = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    for t in (keys2, w3t, b3, scale3, shift3, w4, b4, mean):
        if not t.is_contiguous():
            raise ValueError("pose_tail: contiguous operands expected")
    if tuple(w3t.shape) != (3, 256, 256) or tuple(w4.shape) != (3, 8, 256) or tuple(keys2.shape) != (3, B, 256):
        raise ValueError("pose_tail: shapes")
    check(_lib.lib().tgp_pose_tail(_p(keys2), _p(w3t), _p(b3), _p(scale3), _p(shift3), _p(w4), _p(b4), _p(mean), B, _p(pg), _p(pr),
                                   _p(fg), _p(fr), _p(pT), _p(ps), _p(raw), _stream(keys2)), "tgp_pose_tail")
    return pg, pr, fg, fr, pT, ps


def head_post_bwd(green, red, grads):
    """grads: gradients of head_post's six outputs (None = zero) -> d green (B,4), d red (B,4), d ts (B,6)"""
    B, dev = green.shape[0], green.device
    gs = [None if t is None else t.float().contiguous() for t in grads]
    dg, dr, dt = torch.empty(B, 4, device=dev), torch.empty(B, 4, device=dev), torch.empty(B, 6, device=dev)
    check(_lib.lib().tgp_head_post_bwd(_p(green), _p(red), green.stride(0), red.stride(0), B, *[_p(t) for t in gs], _p(dg), _p(dr),
                                       _p(dt), _stream(green)), "tgp_head_post_bwd")
    return dg, dr, dt


@_timed("graph")
def add_mean_(recon, mean):
    B, n, _ = recon.shape
    check(_lib.lib().tgp_add_mean(_p(recon), _p(mean), B, n, _stream(recon)), "tgp_add_mean")
    return recon


def chamfer_fwd(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_3D.forward convention: caller allocates every output (dist_chamfer_3D.py:33-45)."""
    _f32(xyz1, "xyz1", 3), _f32(xyz2, "xyz2", 3)
    if not (xyz1.is_contiguous() and xyz2.is_contiguous()):
        raise ValueError("chamfer inputs must be contiguous (dist_chamfer_3D.py:72-73)")
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    check(_lib.lib().tgp_chamfer_fwd(_p(xyz1), _p(xyz2), B, n, m, _p(dist1), _p(dist2), _p(_i32(idx1, "idx1")),
                                     _p(_i32(idx2, "idx2")), _stream(xyz1)), "tgp_chamfer_fwd")
    return 1


def chamfer_bwd(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    check(_lib.lib().tgp_chamfer_bwd(_p(xyz1), _p(xyz2), B, n, m, _p(graddist1), _p(graddist2), _p(idx1), _p(idx2),
                                     _p(gradxyz1), _p(gradxyz2), _stream(xyz1)), "tgp_chamfer_bwd")
    return 1


def dcd_fwd(dist1, dist2, idx1, idx2, alpha, n_lambda, non_reg=False, want_weights=True):
    """calc_dcd after the Chamfer search -> (loss (B,), w1 (B,n), w2 (B,m))"""
    B, n = dist1.shape
    m = dist2.shape[1]
    dev = dist1.device
    loss = torch.empty(B, device=dev, dtype=torch.float32)
    w1 = torch.empty(B, n, device=dev, dtype=torch.float32) if want_weights else None
    w2 = torch.empty(B, m, device=dev, dtype=torch.float32) if want_weights else None
    check(_lib.lib().tgp_dcd_fwd(_p(dist1.contiguous()), _p(dist2.contiguous()), _p(_i32(idx1, "idx1")), _p(_i32(idx2, "idx2")),
                                 B, n, m, float(alpha), float(n_lambda), int(bool(non_reg)), _p(loss), _p(w1), _p(w2),
                                 _stream(dist1)), "tgp_dcd_fwd")
    return loss, w1, w2


def dcd_bwd(dist1, dist2, w1, w2, gloss, alpha):
    B, n = dist1.shape
    m = dist2.shape[1]
    gd1, gd2 = torch.empty_like(dist1), torch.empty_like(dist2)
    check(_lib.lib().tgp_dcd_bwd(_p(dist1), _p(dist2), _p(w1), _p(w2), _p(gloss.contiguous()), B, n, m, float(alpha), _p(gd1),
                                 _p(gd2), _stream(dist1)), "tgp_dcd_bwd")
    return gd1, gd2


def canonicalize(points, gR, p_g, f_g, p_r, f_r, p_t, p_s, sym):
    """R_DCD pose normalisation -> (points_re_n (B,n,3), R (B,3,3))"""
    c = lambda t: _f32(t.contiguous(), "arg")
    points, gR, p_g, f_g, p_r, f_r, p_t, p_s, sym = map(c, (points, gR, p_g, f_g, p_r, f_r, p_t, p_s, sym))
    B, n, _ = points.shape
    out = torch.empty_like(points)
    R = torch.empty(B, 3, 3, device=points.device, dtype=torch.float32)
    check(_lib.lib().tgp_canonicalize(_p(points), _p(gR), _p(p_g), _p(f_g), _p(p_r), _p(f_r), _p(p_t), _p(p_s), _p(sym),
                                      sym.shape[1] if sym.dim() > 1 else 1, B, n, _p(out), _p(R), _stream(points)),
          "tgp_canonicalize")
    return out, R


def generate_rt(p_green, p_red, f_green, f_red, T, sym=None):
    """-> (B,4,4) pose matrices"""
    c = lambda t: _f32(t.contiguous(), "arg")
    p_green, p_red, f_green, f_red, T = map(c, (p_green, p_red, f_green.reshape(-1), f_red.reshape(-1), T))
    B = p_green.shape[0]
    out = torch.empty(B, 4, 4, device=p_green.device, dtype=torch.float32)
    sld = 0
    if sym is not None:
        sym = c(sym.float())
        sld = sym.shape[1] if sym.dim() > 1 else 1
    check(_lib.lib().tgp_generate_rt(_p(p_green), _p(p_red), _p(f_green), _p(f_red), _p(T), _p(sym), sld, B, _p(out),
                                     _stream(p_green)), "tgp_generate_rt")
    return out


def bn_train(x, gamma, beta, eps=1e-5, act=0, slope=0.0, slope_vec=None, out=None, colmax_keys=None, cm_cols=0,
             rows_per_obj=0, want_out=True, running=None):
    """Training-mode BatchNorm over rows: x (..., C) rows (row stride may exceed C).  Normalises (in place unless `out`),
    applies the activation, optionally feeds colmax keys.  Returns (out, batch_mean (C,), biased batch_var (C,)).
    running = (running_mean (C,), running_var (C,), momentum, num_batches_tracked or None): the module's buffers, updated by the
    statistics kernel itself (tgp_bn_stats_running) instead of by four torch launches."""
    x, ld = _rows(x, "x")
    C = x.shape[-1]
    rows = math.prod(x.shape[:-1])
    dev = x.device
    mean = torch.empty(C, device=dev, dtype=torch.float32)
    var = torch.empty(C, device=dev, dtype=torch.float32)
    ws = torch.empty(_lib.lib().tgp_bn_workspace_floats(rows, C), device=dev, dtype=torch.float32)
    if running is None:
        check(_lib.lib().tgp_bn_stats(_p(x), ld, rows, C, _p(mean), _p(var), _p(ws), _stream(x)), "tgp_bn_stats")
    else:
        rm, rv, momentum, nbt = running
        if not (rm.is_contiguous() and rv.is_contiguous() and rm.numel() == C and rv.numel() == C and rm.dtype == torch.float32):
            raise ValueError("bn_train: running statistics must be contiguous float32 (C,) tensors")
        if nbt is not None and nbt.dtype != torch.int64:
            raise ValueError("bn_train: num_batches_tracked must be int64")
        check(_lib.lib().tgp_bn_stats_running(_p(x), ld, rows, C, _p(mean), _p(var), _p(ws), _p(rm), _p(rv), float(momentum), _p(nbt),
                                              _stream(x)), "tgp_bn_stats_running")
    ldo = 0
    if want_out:
        out = x if out is None else out
        out, ldo = _rows(out, "out")
    else:
        out = None
    check(_lib.lib().tgp_bn_apply(_p(x), ld, rows, C, _p(mean), _p(var), _p(gamma), _p(beta), float(eps), act, float(slope),
                                  _p(slope_vec), _p(out), ldo, _p(colmax_keys),
                                  colmax_keys.stride(-2) if colmax_keys is not None else 0, cm_cols, rows_per_obj, _stream(x)),
          "tgp_bn_apply")
    return out, mean, var


def dropout(x, p, generator=None):
    """nn.Dropout(p) in train mode: the keep mask comes from torch's generator on the tensor's device."""
    if p <= 0.0:
        return x
    x = x.contiguous()
    keep = (torch.rand(x.shape, device=x.device, generator=generator) >= p).to(torch.uint8)
    y = torch.empty_like(x)
    check(_lib.lib().tgp_dropout_apply(_p(x), _p(keep), float(p), x.numel(), _p(y), _stream(x)), "tgp_dropout_apply")
    return y


# ------------------------------------------------------------------------------------------------- backward pass
def _ws(floats, dev):
    return torch.empty(max(int(floats), 1), device=dev, dtype=torch.float32)


FP16_TOP = 32768.0           # scaled gradients peak in [2^14, 2^15): a factor of two under fp16's largest finite value
TN_SPLIT = os.environ.get("TGP_TN_SPLIT", "1") != "0"    # backward GEMMs of large layers on the scaled fp16 split (0: bf16x3 / fp32 MFMA)
TN_SPLIT_MIN = 128 * 256     # smallest N * K routed to the split path
TN_NATIVE = os.environ.get("TGP_TN_NATIVE", "1") != "0"  # dW of the split path without transposed copies (0: transpose + NT kernel)


def absmax_scale(x, target=FP16_TOP):
    """-> device tensor {s, 1/s, max|x|}: s the largest power of two with max|x| * s <= target (chosen on the device)"""
    x, ld = _rows(x, "x")
    rows, cols = math.prod(x.shape[:-1]), x.shape[-1]
    ws = torch.empty(2048, device=x.device, dtype=torch.int32)
    out = torch.empty(3, device=x.device, dtype=torch.float32)
    check(_lib.lib().tgp_absmax_scale(_p(x), ld, rows, cols, float(target), _p(ws), _p(out), _stream(x)), "tgp_absmax_scale")
    return out


def _ksplit_plan(rows, tiles):
    """number of K-chunks Z and chunk length for a reduction over `rows` feeding `tiles` output tiles: fill whole rounds of the
    256 resident workgroups, keep a chunk >= 512 rows"""
    best, best_eff = 1, 0.0
    for Z in range(1, max(1, rows // 512) + 1):
        t = tiles * Z
        eff = t / (256.0 * ((t + 255) // 256))
        if eff > best_eff + 0.02:
            best, best_eff = Z, eff
        if t >= 1024:
            break
    chunk = ((rows + best - 1) // best + 15) // 16 * 16
    return best, chunk


def gemm_tn_split(a, b, scale, out=None, accumulate=False):
    """gemm_tn on the fp16 split kernels: out (N, K) (+)= a^T b, a = dy (rows, N) scaled into fp16's range by scale[0]
    (absmax_scale(a)), b = x (rows, K) activations.  a^T is transposed + scaled in fp32 and split on the fly as the GEMM's A
    operand, b^T transposed straight into fp16 planes; the reduction over the rows is cut into Z chunks whose partial
    products are added in order and unscaled by scale[1]."""
    a, lda = _rows(a.reshape(-1, a.shape[-1]) if a.dim() > 2 and a.is_contiguous() else a, "a")
    b, ldb = _rows(b.reshape(-1, b.shape[-1]) if b.dim() > 2 and b.is_contiguous() else b, "b")
    rows, N, K = math.prod(a.shape[:-1]), a.shape[-1], b.shape[-1]
    dev = a.device
    Z, chunk = _ksplit_plan(rows, ((N + 255) // 256) * ((K + 255) // 256))
    if TN_NATIVE and N % 4 == 0 and K % 4 == 0 and lda % 4 == 0 and ldb % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0:
        # (round 3) the operands as they lie: transposed LDS reads instead of transposed copies (csrc/gemm_tn_split.hip)
        parts = torch.empty(Z, N, K, device=dev, dtype=torch.float32)
        check(_lib.lib().tgp_gemm_tn_split(_p(a), lda, _p(b), ldb, rows, N, K, _p(scale), Z, chunk, _p(parts), _stream(a)),
              "tgp_gemm_tn_split")
        if out is None:
            out = torch.empty(N, K, device=dev, dtype=torch.float32)
        if not out.is_contiguous():
            raise ValueError("gemm_tn_split: out must be contiguous")
        check(_lib.lib().tgp_sum_slabs(_p(parts), Z, N * K, _p(scale[1:]), _p(out), int(accumulate), _stream(a)), "tgp_sum_slabs")
        return out
    pad = Z * chunk
    at = torch.empty(N, pad, device=dev, dtype=torch.float32)
    check(_lib.lib().tgp_transpose_scaled(_p(a), lda, rows, N, _p(scale), _p(at), pad, _stream(a)), "tgp_transpose_scaled")
    bt = torch.empty(K, pad // 16, 2, 16, device=dev, dtype=torch.int16)
    check(_lib.lib().tgp_transpose_split_f16(_p(b), ldb, rows, K, None, _p(bt), pad, _stream(a)), "tgp_transpose_split_f16")
    parts = torch.empty(Z, N, K, device=dev, dtype=torch.float32)
    gemm(at, at, parts, M=N, N=K, K=chunk, lda=pad, ldw=pad, ldc=K, batch=Z, batch_strides=(chunk, 0, N * K, 0, 0), w_split=bt,
         ksplit_chunk=chunk)
    if out is None:
        out = torch.empty(N, K, device=dev, dtype=torch.float32)
    if not out.is_contiguous():
        raise ValueError("gemm_tn_split: out must be contiguous")
    check(_lib.lib().tgp_sum_slabs(_p(parts), Z, N * K, _p(scale[1:]), _p(out), int(accumulate), _stream(a)), "tgp_sum_slabs")
    return out


def sum_slabs(parts, scale, out=None, accumulate=False):
    """out (+)= scale[0] * sum over the leading dim of parts (Z, ...), slabs added in order (tgp_sum_slabs)"""
    Z = parts.shape[0]
    n = parts.numel() // Z
    if out is None:
        out = torch.empty(parts.shape[1:], device=parts.device, dtype=torch.float32)
    if not out.is_contiguous():
        raise ValueError("sum_slabs: out must be contiguous")
    check(_lib.lib().tgp_sum_slabs(_p(parts), Z, n, _p(scale), _p(out), int(accumulate), _stream(parts)), "tgp_sum_slabs")
    return out


def tn_split_ok(rows, N, K):
    """does gemm_tn route (rows, N)^T (rows, K) to the fp16 split path?  Large enough for the tile kernels in both output dims."""
    if not (TN_SPLIT and GEMM_MODE == "split16" and rows >= 2048 and N * K >= TN_SPLIT_MIN and N > 32 and K > 64):
        return False
    Z, _ = _ksplit_plan(rows, ((N + 255) // 256) * ((K + 255) // 256))
    return _routes_to_big_tile(N, K, Z)


def gemm_tn(a, b, out=None, accumulate=False, scale=None):
    """out (N, K) (+)= a^T b over the rows: a (rows, N), b (rows, K) row views (row stride >= width).  dW = dx^T x.
    scale: absmax_scale(a) if the caller already has it (the same dy feeds the dx GEMM)."""
    a, lda = _rows(a.reshape(-1, a.shape[-1]) if a.dim() > 2 and a.is_contiguous() else a, "a")
    b, ldb = _rows(b.reshape(-1, b.shape[-1]) if b.dim() > 2 and b.is_contiguous() else b, "b")
    rows, N, K = math.prod(a.shape[:-1]), a.shape[-1], b.shape[-1]
    if math.prod(b.shape[:-1]) != rows:
        raise ValueError("gemm_tn: row counts differ")
    if tn_split_ok(rows, N, K) and (out is None or out.is_contiguous()):
        return gemm_tn_split(a, b, absmax_scale(a) if scale is None else scale, out, accumulate)
    if out is None:
        out = torch.empty(N, K, device=a.device, dtype=torch.float32)
    out, ldc = _rows(out, "out")
    ws = _ws(_lib.lib().tgp_gemm_tn_workspace_floats(rows, N, K), a.device)
    check(_lib.lib().tgp_gemm_tn_f32(_p(a), lda, _p(b), ldb, rows, N, K, _p(out), ldc, int(accumulate), _p(ws), _stream(a)),
          "tgp_gemm_tn_f32")
    return out


def colsum(dy, out=None, accumulate=False):
    dy, ld = _rows(dy, "dy")
    rows, C = math.prod(dy.shape[:-1]), dy.shape[-1]
    if out is None:
        out = torch.empty(C, device=dy.device, dtype=torch.float32)
    ws = _ws(_lib.lib().tgp_bw_workspace_floats(rows, C), dy.device)
    check(_lib.lib().tgp_colsum(_p(dy), ld, rows, C, _p(out), int(accumulate), _p(ws), _stream(dy)), "tgp_colsum")
    return out


def bn_bwd(dy, x, mean, var, gamma, beta, eps=1e-5, act=0, slope=0.0, slope_vec=None, dx=None, want_scale=False):
    """Backward of y = act(BatchNorm_train(x)) over rows.  Returns (dx, dgamma, dbeta); dx defaults to in place on dy.
    want_scale: also absmax_scale(dx), collected by the apply pass itself (attached to dx as ``_tgp_scale`` for the linear layer
    whose backward consumes dx next; skipped when the operands do not take the 16-byte form)."""
    dy, lddy = _rows(dy, "dy")
    x, ld = _rows(x, "x")
    rows, C = math.prod(x.shape[:-1]), x.shape[-1]
    dx = dy if dx is None else dx
    dx, lddx = _rows(dx, "dx")
    dg = torch.empty(C, device=x.device, dtype=torch.float32)
    db = torch.empty(C, device=x.device, dtype=torch.float32)
    ws = _ws(_lib.lib().tgp_bw_workspace_floats(rows, C), x.device)
    bits = None
    if want_scale and C % 4 == 0 and lddy % 4 == 0 and ld % 4 == 0 and lddx % 4 == 0 and \
            all(t.data_ptr() % 16 == 0 for t in (dy, x, dx, ws)):
        bits = torch.empty(int(_lib.lib().tgp_bn_bwd_absmax_words(rows, C)), device=x.device, dtype=torch.int32)
    check(_lib.lib().tgp_bn_bwd(_p(dy), lddy, _p(x), ld, rows, C, _p(mean), _p(var), float(eps), _p(gamma), _p(beta), act,
                                float(slope), _p(slope_vec), _p(dx), lddx, _p(dg), _p(db), _p(ws), _p(bits), _stream(x)), "tgp_bn_bwd")
    if bits is not None:
        sc = torch.empty(3, device=x.device, dtype=torch.float32)
        check(_lib.lib().tgp_absmax_scale_from_bits(_p(bits), bits.numel(), float(FP16_TOP), _p(sc), _stream(x)),
              "tgp_absmax_scale_from_bits")
        dx._tgp_scale = sc
    return dx, dg, db


def bn_bwd_pooled(dpool, argrow, x, rows_per_obj, mean, var, gamma, beta, eps=1e-5, act=0, slope=0.0, slope_vec=None, dx=None):
    """Backward of max-over-points(act(BatchNorm_train(x))): dpool (objects, C), argrow (objects, C) int32 -> dense dx."""
    x, ld = _rows(x, "x")
    objects, C = dpool.shape
    if dx is None:
        dx = torch.empty(objects * rows_per_obj, C, device=x.device, dtype=torch.float32)
    dx, lddx = _rows(dx, "dx")
    dg = torch.empty(C, device=x.device, dtype=torch.float32)
    db = torch.empty(C, device=x.device, dtype=torch.float32)
    check(_lib.lib().tgp_bn_bwd_pooled(_p(dpool), dpool.stride(0), _p(argrow), argrow.stride(0), _p(x), ld, objects, rows_per_obj,
                                       C, _p(mean), _p(var), float(eps), _p(gamma), _p(beta), act, float(slope), _p(slope_vec),
                                       _p(dx), lddx, _p(dg), _p(db), _stream(x)), "tgp_bn_bwd_pooled")
    return dx, dg, db


def colmax_arg(x, objects, n, bn=None, act=0, slope=0.0, slope_vec=None, eps=1e-5):
    """max over each object's n rows (+ winning global row, int32) of act(BN(x)); bn = (mean, var, gamma, beta) or None."""
    x, ld = _rows(x, "x")
    C = x.shape[-1]
    out = torch.empty(objects, C, device=x.device, dtype=torch.float32)
    arg = torch.empty(objects, C, device=x.device, dtype=torch.int32)
    mean, var, gamma, beta = bn if bn is not None else (None, None, None, None)
    check(_lib.lib().tgp_colmax_arg(_p(x), ld, objects, n, C, _p(mean), _p(var), float(eps), _p(gamma), _p(beta), act,
                                    float(slope), _p(slope_vec), _p(out), out.stride(0), _p(arg), arg.stride(0), _stream(x)),
          "tgp_colmax_arg")
    return out, arg


def colsum_objects(dy):
    """dy (B, n, C) rows -> (B, C): sum over each object's points, fixed order"""
    dy, ld = _rows(dy, "dy")
    B, n, C = dy.shape
    out = torch.empty(B, C, device=dy.device, dtype=torch.float32)
    check(_lib.lib().tgp_colsum_objects(_p(dy), ld, B, n, C, _p(out), C, _stream(dy)), "tgp_colsum_objects")
    return out


def colmax_bwd(dpool, argrow, rows_per_obj, dx=None):
    """backward of colmax_arg(x) without BatchNorm: dpool (objects, C), argrow (objects, C) int32 -> dense dx (objects * n, C)"""
    objects, C = dpool.shape
    if dx is None:
        dx = torch.empty(objects * rows_per_obj, C, device=dpool.device, dtype=torch.float32)
    dx, lddx = _rows(dx, "dx")
    check(_lib.lib().tgp_colmax_bwd(_p(dpool), dpool.stride(0), _p(argrow), argrow.stride(0), objects, rows_per_obj, C, _p(dx), lddx,
                                    _stream(dpool)), "tgp_colmax_bwd")
    return dx


def transpose(w):
    """(rows, cols) -> contiguous (cols, rows)"""
    w, ld = _rows(w, "w")
    rows, cols = w.shape
    out = torch.empty(cols, rows, device=w.device, dtype=torch.float32)
    check(_lib.lib().tgp_transpose(_p(w), ld, rows, cols, _p(out), rows, _stream(w)), "tgp_transpose")
    return out


def transpose_both(w):
    """w (rows, cols) -> (wt (cols, rows16) fp32 = w^T zero padded to a multiple of 16 columns, split_f16(wt)) in one launch"""
    w, ld = _rows(w, "w")
    rows, cols = w.shape
    pad = (rows + 15) // 16 * 16
    wt = torch.empty(cols, pad, device=w.device, dtype=torch.float32)
    planes = torch.empty(cols, pad // 16, 2, 16, device=w.device, dtype=torch.int16)
    check(_lib.lib().tgp_transpose_both(_p(w), ld, rows, cols, _p(wt), _p(planes), pad, _stream(w)), "tgp_transpose_both")
    return wt, planes


def gconv_surface_bwd(xyz, idx, sdn, dg, S, C):
    """-> dsdn (3, S*C)"""
    dg, ldg = _rows(dg, "dg")
    B, n, k = idx.shape
    dsdn = torch.empty(3, S * C, device=xyz.device, dtype=torch.float32)
    ws = _ws(_lib.lib().tgp_gconv_bwd_workspace_floats(B, n, C), xyz.device)
    check(_lib.lib().tgp_gconv_surface_bwd(_p(xyz), _p(idx), _p(sdn), _p(dg), ldg, B, n, k, S, C, _p(dsdn), _p(ws), _stream(xyz)),
          "tgp_gconv_surface_bwd")
    return dsdn


def gconv_hs_bwd(xyz, idx, proj, sdn, dg, S, C):
    """-> (dproj (B,n,8C) = [d centre | d support], dsdn (3, S*C))"""
    proj, ldp = _rows(proj, "proj")
    dg, ldg = _rows(dg, "dg")
    B, n, k = idx.shape
    dproj = torch.zeros(B, n, 8 * C, device=xyz.device, dtype=torch.float32)
    dsdn = torch.empty(3, S * C, device=xyz.device, dtype=torch.float32)
    ws = _ws(_lib.lib().tgp_gconv_bwd_workspace_floats(B, n, C), xyz.device)
    check(_lib.lib().tgp_gconv_hs_bwd(_p(xyz), _p(idx), _p(proj), ldp, _p(sdn), _p(dg), ldg, B, n, k, S, C, _p(dproj), 8 * C,
                                      _p(dsdn), _p(ws), _stream(xyz)), "tgp_gconv_hs_bwd")
    return dproj, dsdn


def nbrmax_bwd(src, idx, dy, per_object=False, scale=1.0, dsrc=None):
    """backward of y[b,p] = max_j src[b, idx[b,p,j]]: idx (B, n_rows, k) int32; dy (B,n_rows,C) or (B,C) when per_object."""
    src, lds = _rows(src, "src")
    _i32(idx, "idx")
    B, n_src, C = src.shape
    n_rows, k = idx.shape[1], idx.shape[2]
    dy, lddy = _rows(dy, "dy")
    if dsrc is None:
        dsrc = torch.zeros(B, n_src, C, device=src.device, dtype=torch.float32)
    dsrc, ldds = _rows(dsrc, "dsrc")
    check(_lib.lib().tgp_nbrmax_bwd(_p(src), lds, _p(idx), B, n_src, n_rows, k, C, _p(dy), lddy, int(per_object), float(scale),
                                    _p(dsrc), ldds, _stream(src)), "tgp_nbrmax_bwd")
    return dsrc


def gather_rows_bwd(dy, idx, n_src):
    """backward of gather_rows: dy (B,n_out,C) rows, idx (B,n_out) int32 -> dsrc (B,n_src,C)"""
    dy, lddy = _rows(dy, "dy")
    _i32(idx, "idx")
    B, n_out, C = dy.shape
    dsrc = torch.zeros(B, n_src, C, device=dy.device, dtype=torch.float32)
    check(_lib.lib().tgp_gather_rows_bwd(_p(dy), lddy, _p(idx), B, n_src, n_out, C, _p(dsrc), C, _stream(dy)),
          "tgp_gather_rows_bwd")
    return dsrc


def pose_rotation_fwd(gR0, p_g, f_g, p_r, f_r, sym0):
    """-> R (B,3,3), J (B,9,8): tgp_pose_rotation_fwd (all inputs float32, contiguous, on the device)"""
    B = p_g.shape[0]
    R = torch.empty(B, 3, 3, device=p_g.device, dtype=torch.float32)
    J = torch.empty(B, 9, 8, device=p_g.device, dtype=torch.float32)
    check(_lib.lib().tgp_pose_rotation_fwd(_p(gR0), _p(p_g), _p(f_g), _p(p_r), _p(f_r), _p(sym0), B, _p(R), _p(J), _stream(p_g)),
          "tgp_pose_rotation_fwd")
    return R, J


def pose_rotation_bwd(dR, J):
    B = J.shape[0]
    din = torch.empty(B, 8, device=J.device, dtype=torch.float32)
    check(_lib.lib().tgp_pose_rotation_bwd(_p(dR), _p(J), B, _p(din), _stream(J)), "tgp_pose_rotation_bwd")
    return din


def reverse_graph(idx, n_src):
    """idx (B, n_rows, k) int32 ids in [0, n_src) -> (rptr (B*n_src + 1,), rent (B*n_rows*k,)) int32: for every source row the
    ascending (row << 6 | slot) pairs that list it; None when the shape is outside what tgp_reverse_graph sorts in LDS"""
    _i32(idx, "idx")
    idx = idx.contiguous()
    B, n_rows, k = idx.shape
    if k > 64 or n_src > 8192 or (n_rows * k + 2 * n_src) * 4 > 150 * 1024:
        return None
    rptr = torch.empty(B * n_src + 1, device=idx.device, dtype=torch.int32)
    rent = torch.empty(B * n_rows * k, device=idx.device, dtype=torch.int32)
    check(_lib.lib().tgp_reverse_graph(_p(idx), B, n_rows, k, n_src, _p(rptr), _p(rent), _stream(idx)), "tgp_reverse_graph")
    return rptr, rent


def nbrmax_gather_ok(C, *tensors):
    lanes = C // 4
    return C % 4 == 0 and 0 < lanes <= 256 and 256 % lanes == 0 and all(t.data_ptr() % 16 == 0 for t in tensors)


@_timed("graph")
def nbrmax_bwd_gather(src, idx, rev, dy, per_object=False, scale=1.0):
    """nbrmax_bwd without atomics (rev = reverse_graph(idx, n_src)); dsrc is written densely"""
    src, lds = _rows(src, "src")
    _i32(idx, "idx")
    B, n_src, C = src.shape
    n_rows, k = idx.shape[1], idx.shape[2]
    dy, lddy = _rows(dy, "dy")
    dsrc = torch.empty(B, n_src, C, device=src.device, dtype=torch.float32)
    arg = torch.empty(B * n_rows * C, device=src.device, dtype=torch.uint8)
    check(_lib.lib().tgp_nbrmax_bwd_gather(_p(src), lds, _p(idx), _p(rev[0]), _p(rev[1]), B, n_src, n_rows, k, C, _p(dy), lddy,
                                           int(per_object), float(scale), _p(arg), _p(dsrc), C, _stream(src)), "tgp_nbrmax_bwd_gather")
    return dsrc


def gconv_gather_ok(C, k, *tensors):
    return C in (128, 256, 512) and k <= 63 and all(t.data_ptr() % 16 == 0 for t in tensors)


@_timed("graph")
def gconv_hs_slots(xyz, idx, proj, sdn, S, C):
    """gconv_hs that also records the winning slots for gconv_hs_bwd_gather(slots=...): -> (out (B,n,C), slots uint8 (B*n*S*C,))"""
    _f32(xyz, "xyz", 3), _i32(idx, "idx")
    proj, ldp = _rows(proj, "proj")
    B, n, k = idx.shape
    out = torch.empty(B, n, C, device=xyz.device, dtype=torch.float32)
    slots = torch.empty(B * n * S * C, device=xyz.device, dtype=torch.uint8)
    check(_lib.lib().tgp_gconv_hs_fwd_slots(_p(xyz), _p(idx), _p(proj), ldp, _p(sdn), B, n, k, S, C, _p(out), C, _p(slots),
                                            _stream(xyz)), "tgp_gconv_hs_fwd_slots")
    return out, slots


@_timed("graph")
def gconv_hs_bwd_gather(xyz, idx, rev, proj, sdn, dg, S, C, slots=None):
    """gconv_hs_bwd without atomics (rev = reverse_graph(idx, n)) -> (dproj (B,n,8C), dsdn (3, S*C)).  slots: as recorded by
    gconv_hs_slots in the forward (CONSUMED: entries that carry no gradient are rewritten); None: recomputed here."""
    proj, ldp = _rows(proj, "proj")
    dg, ldg = _rows(dg, "dg")
    B, n, k = idx.shape
    dproj = torch.empty(B, n, 8 * C, device=xyz.device, dtype=torch.float32)
    dsdn = torch.empty(3, S * C, device=xyz.device, dtype=torch.float32)
    ws = _ws(_lib.lib().tgp_gconv_bwd_workspace_floats(B, n, C), xyz.device)
    arg = slots if slots is not None else torch.empty(B * n * S * C, device=xyz.device, dtype=torch.uint8)
    contrib = torch.empty(B * n * S * C, device=xyz.device, dtype=torch.float32)
    check(_lib.lib().tgp_gconv_hs_bwd_gather(_p(xyz), _p(idx), _p(rev[0]), _p(rev[1]), _p(proj), ldp, _p(sdn), _p(dg), ldg, B, n, k, S, C,
                                             _p(dproj), 8 * C, _p(dsdn), _p(ws), _p(arg), _p(contrib), int(slots is not None),
                                             _stream(xyz)), "tgp_gconv_hs_bwd_gather")
    return dproj, dsdn


def child_lists(near, R, global_ids=False):
    """near (B, n) int32 parent of every point, in [0, R) (global_ids: b * R + that) -> (ptr (B*R + 1,), idx (B*n,)) int32: CSR
    child lists over the B*R global parent rows, idx = global point rows b*n + i, children in point order"""
    _i32(near, "near")
    near = near.contiguous()
    B, n = near.shape
    ptr = torch.empty(B * R + 1, device=near.device, dtype=torch.int32)
    idx = torch.empty(B * n, device=near.device, dtype=torch.int32)
    check(_lib.lib().tgp_child_lists(_p(near), B, n, R, 1 if global_ids else 0, _p(ptr), _p(idx), _stream(near)), "tgp_child_lists")
    return ptr, idx


@_timed("graph")
def segsum_rows(g, ptr, idx, out=None):
    """g (M, C) rows -> out (R, C), R = ptr.numel() - 1: out[r] = sum of g[idx[k]] over k in [ptr[r], ptr[r+1]); out may be a
    column slice of a wider buffer"""
    g, ldg = _rows(g, "g")
    C = g.shape[-1]
    R = ptr.numel() - 1
    if out is None:
        out = torch.empty(R, C, device=g.device, dtype=torch.float32)
    out, ldo = _rows(out, "out")
    if tuple(out.shape) != (R, C):
        raise ValueError("segsum_rows: out must be (%d, %d)" % (R, C))
    check(_lib.lib().tgp_segsum_rows(_p(g), ldg, C, _p(ptr), _p(idx), R, _p(out), ldo, _stream(g)), "tgp_segsum_rows")
    return out


def pose_transform(points, R, t, s):
    """out = (R^T (points - t)) * s per object: points (B,n,3), R (B,3,3), t, s (B,3)"""
    points, R, t, s = points.contiguous(), R.contiguous(), t.contiguous(), s.contiguous()
    B, n, _ = points.shape
    out = torch.empty_like(points)
    check(_lib.lib().tgp_pose_transform_fwd(_p(points), _p(R), _p(t), _p(s), B, n, _p(out), _stream(points)), "tgp_pose_transform_fwd")
    return out


def pose_transform_bwd(points, R, t, s, dout, need_points=True):
    points, R, t, s, dout = points.contiguous(), R.contiguous(), t.contiguous(), s.contiguous(), dout.contiguous()
    B, n, _ = points.shape
    dev = points.device
    dp = torch.empty_like(points) if need_points else None
    dR = torch.empty(B, 3, 3, device=dev, dtype=torch.float32)
    dt = torch.empty(B, 3, device=dev, dtype=torch.float32)
    ds = torch.empty(B, 3, device=dev, dtype=torch.float32)
    check(_lib.lib().tgp_pose_transform_bwd(_p(points), _p(R), _p(t), _p(s), _p(dout), B, n, _p(dp), _p(dR), _p(dt), _p(ds),
                                            _stream(points)), "tgp_pose_transform_bwd")
    return dp, dR, dt, ds


# ---- the regression terms of the training loss (csrc/tdaloss.hip) ---------------------------------------------------------

def sym_i32(sym):
    """sym_info as the kernels read it: (B, S) int32, contiguous"""
    if sym.dim() != 2:
        raise ValueError("sym must be (B, S)")
    return sym.to(torch.int32).contiguous()


def _pose_args(pred, gt, sym, kind, beta):
    B = pred[0].shape[0]
    for t in tuple(pred) + tuple(gt):
        _f32(t, "pose term operand")
        if t.shape[0] != B or not t.is_contiguous():
            raise ValueError("pose term operands must be contiguous with %d rows" % B)
    if sym.dtype != torch.int32 or sym.shape[0] != B or not sym.is_contiguous():
        raise ValueError("sym must be (B, S) int32 contiguous (ops.sym_i32)")
    return [_p(t) for t in pred] + [_p(t) for t in gt] + [_p(sym), sym.shape[1], B, int(kind), float(beta)]


def pose_terms_fwd(pred, gt, sym, kind=0, beta=0.5):
    """pred = (rot1, rot2, f1, f2, tran, size), gt = (rot1, rot2, tran, size) -> 9 floats: the eight unweighted terms + valid count"""
    out = torch.empty(9, device=pred[0].device, dtype=torch.float32)
    check(_lib.lib().tgp_pose_terms_fwd(*_pose_args(pred, gt, sym, kind, beta), _p(out), _stream(out)), "tgp_pose_terms_fwd")
    return out


def pose_terms_bwd(pred, gt, sym, kind, beta, fwd_out, gw):
    grads = [torch.empty_like(t) for t in pred]
    d = dict(zip(("rot1", "rot2", "f1", "f2", "tran", "size"), grads))
    check(_lib.lib().tgp_pose_terms_bwd(*_pose_args(pred, gt, sym, kind, beta), _p(fwd_out), _p(gw), _p(d["rot1"]), _p(d["rot2"]),
                                        _p(d["f1"]), _p(d["f2"]), _p(d["tran"]), _p(d["size"]), _stream(fwd_out)), "tgp_pose_terms_bwd")
    return grads


def _sym_recon_args(PC, PC_re, gt_R, gt_t, sym):
    for t, n in ((PC, "PC"), (PC_re, "PC_re"), (gt_R, "gt_R"), (gt_t, "gt_t")):
        _f32(t, n)
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % n)
    B, N, _ = PC.shape
    if PC_re.shape != PC.shape or gt_R.shape != (B, 3, 3) or gt_t.shape != (B, 3) or sym.shape[0] != B or sym.dtype != torch.int32:
        raise ValueError("prop_sym_matching_loss operand shapes")
    return [_p(PC), _p(PC_re), _p(gt_R), _p(gt_t), _p(sym), sym.shape[1], sym.shape[1], B, N]


def sym_recon_fwd(PC, PC_re, gt_R, gt_t, sym):
    B, N, _ = PC.shape
    ws = torch.empty(int(_lib.lib().tgp_sym_recon_workspace_floats(B, N)), device=PC.device, dtype=torch.float32)
    loss = torch.empty(1, device=PC.device, dtype=torch.float32)
    check(_lib.lib().tgp_sym_recon_fwd(*_sym_recon_args(PC, PC_re, gt_R, gt_t, sym), _p(ws), _p(loss), _stream(PC)), "tgp_sym_recon_fwd")
    return loss


def sym_recon_bwd(PC, PC_re, gt_R, gt_t, sym, gloss, need_pc, need_re):
    dPC = torch.empty_like(PC) if need_pc else None
    dRe = torch.empty_like(PC_re) if need_re else None
    check(_lib.lib().tgp_sym_recon_bwd(*_sym_recon_args(PC, PC_re, gt_R, gt_t, sym), _p(gloss), _p(dPC), _p(dRe), _stream(PC)),
          "tgp_sym_recon_bwd")
    return dPC, dRe


def rowl1_fwd(a, b, wsrc):
    for t in (a, b, wsrc):
        _f32(t, "rowl1 operand", 2)
    if a.shape != b.shape or a.shape != wsrc.shape or not (a.is_contiguous() and b.is_contiguous() and wsrc.is_contiguous()):
        raise ValueError("rowl1 operands must be contiguous and of one shape")
    B, D = a.shape
    rows = torch.empty(B, 4, device=a.device, dtype=torch.float32)
    out = torch.empty(2, device=a.device, dtype=torch.float32)
    check(_lib.lib().tgp_rowl1_fwd(_p(a), _p(b), _p(wsrc), B, D, _p(rows), _p(out), _stream(a)), "tgp_rowl1_fwd")
    return out, rows


def rowl1_bwd(a, b, rows, fwd_out, gout):
    B, D = a.shape
    da = torch.empty_like(a)
    check(_lib.lib().tgp_rowl1_bwd(_p(a), _p(b), _p(rows), _p(fwd_out), _p(gout), B, D, _p(da), _stream(a)), "tgp_rowl1_bwd")
    return da


def feat_consistency_fwd(x1, x2):
    _f32(x1, "x1", 2), _f32(x2, "x2", 2)
    if x1.shape != x2.shape or not (x1.is_contiguous() and x2.is_contiguous()):
        raise ValueError("feat_consistency operands must be contiguous and of one shape")
    B, C = x1.shape
    rows = torch.empty(B, 3, device=x1.device, dtype=torch.float32)
    loss = torch.empty(1, device=x1.device, dtype=torch.float32)
    check(_lib.lib().tgp_feat_consistency_fwd(_p(x1), _p(x2), B, C, _p(rows), _p(loss), _stream(x1)), "tgp_feat_consistency_fwd")
    return loss, rows


def feat_consistency_bwd(x1, x2, rows, gloss, need1, need2):
    B, C = x1.shape
    d1 = torch.empty_like(x1) if need1 else None
    d2 = torch.empty_like(x2) if need2 else None
    check(_lib.lib().tgp_feat_consistency_bwd(_p(x1), _p(x2), _p(rows), _p(gloss), B, C, _p(d1), _p(d2), _stream(x1)),
          "tgp_feat_consistency_bwd")
    return d1, d2


# ----------------------------------------------------------------------------- input side (depth image -> cloud)
class RoiRecords:
    """What tgp_roi_cloud leaves in HBM: ``recs`` (D, roi^2) int32-viewed records (ROI pixel index << 16 | depth), ``counts``
    (D,3) int32, and the descriptor tensors a record needs to become a point (``det_img``, ``window``, ``camk``)."""

    def __init__(self, recs, counts, det_img, window, camk, roi_size, tables=None):
        self.recs, self.counts, self.det_img, self.window, self.camk, self.roi_size = recs, counts, det_img, window, camk, roi_size
        self.tables = tables


def roi_cloud(depth, masks, mask_off, mask_stride, det_img, window, camk, roi_size=256, tables=None, mask_val=None, cut_frac=0.25):
    """tgp_roi_cloud / tgp_roi_cloud_ex.  depth (I,H,W) int16-viewed uint16, masks flat uint8, mask_off (D,) int64, mask_stride /
    det_img (D,) int32, window (D,3) int32 (None with tables), camk (I,4) float32 -- all on the GPU.  tables (D,2,roi_size) int32:
    host-evaluated source pixels (the training loader's augmented windows); mask_val (D,) int32: mask byte to match (ground-truth
    instance masks); cut_frac: the outlier cut's share of the diagonal (0.25 evaluation, 0.15 training).  -> RoiRecords"""
    if not (depth.is_cuda and depth.dtype in (torch.int16, torch.uint16) and depth.dim() == 3 and depth.is_contiguous()):
        raise TypeError("depth must be a contiguous (I,H,W) 16-bit GPU tensor")
    if not (masks.is_cuda and masks.dtype in (torch.uint8, torch.bool) and masks.is_contiguous()):
        raise TypeError("masks must be a contiguous uint8 / bool GPU tensor")
    if not (mask_off.is_cuda and mask_off.dtype == torch.int64 and mask_off.is_contiguous()):
        raise TypeError("mask_off must be a contiguous int64 GPU tensor")
    _i32(mask_stride, "mask_stride"), _i32(det_img, "det_img")
    _f32(camk, "camk", 2)
    I, H, W = depth.shape
    D = det_img.numel()
    if window is None and tables is None:
        raise ValueError("roi_cloud: a window or source tables are needed")
    if window is not None and _i32(window, "window").shape != (D, 3):
        raise ValueError("roi_cloud: window must be (D,3)")
    if tables is not None and _i32(tables, "tables").shape != (D, 2, roi_size):
        raise ValueError("roi_cloud: tables must be (D,2,roi_size)")
    if mask_val is not None and _i32(mask_val, "mask_val").numel() != D:
        raise ValueError("roi_cloud: mask_val must be (D,)")
    if mask_off.numel() != D or mask_stride.numel() != D or camk.shape != (I, 4) or not camk.is_contiguous():
        raise ValueError("roi_cloud: inconsistent shapes")
    recs = torch.empty(D, roi_size * roi_size, device=depth.device, dtype=torch.int32)
    counts = torch.empty(D, 3, device=depth.device, dtype=torch.int32)
    if tables is None and mask_val is None and cut_frac == 0.25:
        check(_lib.lib().tgp_roi_cloud(_p(depth), _p(masks), _p(mask_off), _p(mask_stride), _p(det_img), _p(window), _p(camk), D, H, W,
                                       roi_size, _p(recs), _p(counts), _stream(depth)), "tgp_roi_cloud")
    else:
        check(_lib.lib().tgp_roi_cloud_ex(_p(depth), _p(masks), _p(mask_off), _p(mask_stride), _p(det_img), _p(window), _p(camk), D, H, W,
                                          roi_size, _p(recs), _p(counts), _p(tables), _p(mask_val), float(cut_frac), _stream(depth)),
              "tgp_roi_cloud_ex")
    return RoiRecords(recs, counts, det_img, window, camk, roi_size, tables)


def cloud_select(rr, sel):
    """out[d, i] = point(recs[d, sel[d, i]]); rr RoiRecords, sel (D,n_pts) int32 -> (D,n_pts,3) float32"""
    _i32(sel, "sel")
    D = rr.recs.shape[0]
    if sel.dim() != 2 or sel.shape[0] != D:
        raise ValueError("cloud_select: sel must be (D,n_pts)")
    out = torch.empty(D, sel.shape[1], 3, device=rr.recs.device, dtype=torch.float32)
    if getattr(rr, "tables", None) is None:
        check(_lib.lib().tgp_cloud_select(_p(rr.recs), _p(sel), _p(rr.det_img), _p(rr.window), _p(rr.camk), D, rr.roi_size, sel.shape[1],
                                          _p(out), _stream(rr.recs)), "tgp_cloud_select")
    else:
        check(_lib.lib().tgp_cloud_select_ex(_p(rr.recs), _p(sel), _p(rr.det_img), _p(rr.window), _p(rr.camk), D, rr.roi_size,
                                             sel.shape[1], _p(out), _p(rr.tables), _stream(rr.recs)), "tgp_cloud_select_ex")
    return out


def cloud_sample(rr, n_pts, seed, counts=None):
    """Device-drawn resampling (tgp_cloud_sample) -> (D,n_pts,3); ``counts`` overrides rr.counts (tests)."""
    counts = rr.counts if counts is None else _i32(counts, "counts")
    D = rr.recs.shape[0]
    if getattr(rr, "tables", None) is not None:
        raise ValueError("cloud_sample: records built from source tables are resampled with cloud_select (tgp_cloud_select_ex)")
    if counts.shape != (D, 3):
        raise ValueError("cloud_sample: counts must be (D,3)")
    out = torch.empty(D, n_pts, 3, device=rr.recs.device, dtype=torch.float32)
    check(_lib.lib().tgp_cloud_sample(_p(rr.recs), _p(counts), _p(rr.det_img), _p(rr.window), _p(rr.camk), D, rr.roi_size, n_pts,
                                      int(seed) & (2 ** 64 - 1), _p(out), _stream(rr.recs)), "tgp_cloud_sample")
    return out
