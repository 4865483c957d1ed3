"""ctypes loader for oracle/_build/libtgp_oracle.so (built by oracle/Makefile; test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtgp_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "csrc", "tgp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.tgpo_sqnorm.argtypes = [_f32p, ctypes.c_long, ctypes.c_int, _f32p]
        _lib.tgpo_knn.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p, _f32p, _i32p]
        _lib.tgpo_nn1.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p]
        _lib.tgpo_chamfer_fwd.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _i32p, _i32p]
        _lib.tgpo_chamfer_bwd.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _i32p, _i32p, _f32p, _f32p]
        _lib.tgpo_knn_dist_matrix.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _f32p]
        for f in (_lib.tgpo_sqnorm, _lib.tgpo_knn, _lib.tgpo_nn1, _lib.tgpo_chamfer_fwd, _lib.tgpo_chamfer_bwd,
                  _lib.tgpo_knn_dist_matrix):
            f.restype = None
    return _lib


def f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def sqnorm(x):
    x, px = f32(x)
    rows, d = int(np.prod(x.shape[:-1])), x.shape[-1]
    q = np.empty(x.shape[:-1], np.float32)
    lib().tgpo_sqnorm(px, rows, d, q.ctypes.data_as(_f32p))
    return q


def knn(x, k, want_dist=False, want_first=False):
    """x (B,n,d) float32 -> idx (B,n,k) int32 [, dist (B,n,k)] [, first (B,n)]"""
    x, px = f32(x)
    B, n, d = x.shape
    idx = np.empty((B, n, k), np.int32)
    dist = np.empty((B, n, k), np.float32) if want_dist else None
    first = np.empty((B, n), np.int32) if want_first else None
    lib().tgpo_knn(px, B, n, d, k, idx.ctypes.data_as(_i32p),
                   dist.ctypes.data_as(_f32p) if want_dist else None,
                   first.ctypes.data_as(_i32p) if want_first else None)
    out = (idx,)
    if want_dist:
        out += (dist,)
    if want_first:
        out += (first,)
    return out if len(out) > 1 else idx


def nn1(tgt, src):
    tgt, pt = f32(tgt)
    src, ps = f32(src)
    B, n, d = tgt.shape
    m = src.shape[1]
    idx = np.empty((B, n), np.int32)
    lib().tgpo_nn1(pt, ps, B, n, m, d, idx.ctypes.data_as(_i32p))
    return idx


def chamfer_fwd(xyz1, xyz2):
    xyz1, p1 = f32(xyz1)
    xyz2, p2 = f32(xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((B, n), np.float32)
    d2 = np.empty((B, m), np.float32)
    i1 = np.empty((B, n), np.int32)
    i2 = np.empty((B, m), np.int32)
    lib().tgpo_chamfer_fwd(p1, p2, B, n, m, d1.ctypes.data_as(_f32p), d2.ctypes.data_as(_f32p),
                           i1.ctypes.data_as(_i32p), i2.ctypes.data_as(_i32p))
    return d1, d2, i1, i2


def chamfer_bwd(xyz1, xyz2, gd1, gd2, idx1, idx2):
    xyz1, p1 = f32(xyz1)
    xyz2, p2 = f32(xyz2)
    gd1, pg1 = f32(gd1)
    gd2, pg2 = f32(gd2)
    idx1, pi1 = i32(idx1)
    idx2, pi2 = i32(idx2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros((B, n, 3), np.float32)
    g2 = np.zeros((B, m, 3), np.float32)
    lib().tgpo_chamfer_bwd(p1, p2, B, n, m, pg1, pg2, pi1, pi2, g1.ctypes.data_as(_f32p), g2.ctypes.data_as(_f32p))
    return g1, g2


def knn_dist_matrix(x):
    x, px = f32(x)
    n, d = x.shape
    D = np.empty((n, n), np.float32)
    lib().tgpo_knn_dist_matrix(px, n, d, D.ctypes.data_as(_f32p))
    return D
