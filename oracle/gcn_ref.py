"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the graph operators of the reference's
``network/fs_net_repo/gcn3d.py`` as stateless functions over a parameter dict.

Float work uses the same torch CPU ops the reference uses (so it agrees with the imported reference
to rounding); index work has two back-ends:

* ``mode='exact'``  -- oracle/csrc/tgp_oracle.c: the distance arithmetic pinned bit by bit
  (ascending-k FMA chain, ATen cascade sum, three separate roundings) with (distance, index)
  ordering.  This is what the HIP kernels are compared against, bit-exact.
* ``mode='torch'``  -- the reference's own op sequence (bmm / sum / topk, gcn3d.py:18-22), used to
  cross-check 'exact' and as the timed CPU baseline.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import _clib


# ----------------------------------------------------------------------------- index builders
def knn_index(x, k, mode="exact"):
    """gcn3d.py:14-23 get_neighbor_index.  x (B,n,d) -> int64 (B,n,k), self (rank 0) dropped."""
    if mode == "torch":
        inner = torch.bmm(x, x.transpose(1, 2))
        sq = torch.sum(x ** 2, dim=2)
        dist = inner * (-2) + sq.unsqueeze(1) + sq.unsqueeze(2)
        return torch.topk(dist, k=k + 1, dim=-1, largest=False)[1][:, :, 1:]
    idx = _clib.knn(x.detach().cpu().numpy(), int(k))
    return torch.from_numpy(idx.astype(np.int64))


def nearest_index(target, source, mode="exact"):
    """gcn3d.py:26-35 get_nearest_index.  (B,n,3),(B,m,3) -> int64 (B,n,1)."""
    if mode == "torch":
        inner = torch.bmm(target, source.transpose(1, 2))
        s2 = torch.sum(source ** 2, dim=2)
        t2 = torch.sum(target ** 2, dim=2)
        d = s2.unsqueeze(1) + t2.unsqueeze(2) - 2 * inner
        return torch.topk(d, k=1, dim=-1, largest=False)[1]
    idx = _clib.nn1(target.detach().cpu().numpy(), source.detach().cpu().numpy())
    return torch.from_numpy(idx.astype(np.int64)).unsqueeze(-1)


def gather_rows(t, index):
    """gcn3d.py:38-46 indexing_neighbor_new.  t (B,n,C), index (B,m,k) -> (B,m,k,C)."""
    B, n, C = t.shape
    flat = (index + torch.arange(B).view(B, 1, 1) * n).reshape(-1)
    return t.reshape(B * n, C)[flat].view(B, index.shape[1], index.shape[2], C)


def neighbor_directions(xyz, index):
    """gcn3d.py:48-58: unit vectors from each point to its neighbours (F.normalize eps 1e-12)."""
    return F.normalize(gather_rows(xyz, index) - xyz.unsqueeze(2), dim=-1)


class GraphCache(object):
    """Records every index tensor built during one forward so tests can compare / re-inject them.

    ``inject`` (dict name -> int64 tensor) overrides the computed indices: used for the
    'teacher-forced' end-to-end comparison where both sides must walk the same graph.
    """

    def __init__(self, mode="exact", inject=None):
        self.mode = mode
        self.inject = inject or {}
        self.record = {}

    def knn(self, name, x, k):
        if name in self.inject:
            idx = self.inject[name]
        else:
            idx = knn_index(x, k, self.mode)
        self.record[name] = idx
        return idx

    def nn1(self, name, target, source):
        if name in self.inject:
            idx = self.inject[name]
        else:
            idx = nearest_index(target, source, self.mode)
        self.record[name] = idx
        return idx


# ----------------------------------------------------------------------------- layers
def tapv(P, name, t):
    """(tests: decision forcing, posenet_ref.posenet_forward(force=...)) the value of intermediate `name` replaced by the recorded one,
    gradients flowing through as if it were this tensor"""
    if P.get("_record") is not None:
        P["_record"][name] = t.detach().clone()
    f = P.get("_force")
    if f is None or name not in f:
        return t
    return t + (f[name].to(t.dtype).view(t.shape) - t).detach()


def _pointwise(x, w):
    """Conv1d(kernel 1, no bias) on channel-last rows: x (B,n,Cin), w (Cout,Cin,1) -> (B,n,Cout)."""
    return F.conv1d(x.transpose(1, 2), w).transpose(1, 2).contiguous()


def orl_global(feature, xyz, k, cache, tag):
    """gcn3d.py:210-217 get_ORL_global: neighbour max then mean over all points -> (B,1,C)."""
    idx = cache.knn(tag + ".orl_xyz", xyz, k)
    pooled = gather_rows(feature, idx).max(dim=2)[0]
    return pooled.mean(dim=1, keepdim=True)


def _orl_forward(P, name, feature, xyz, k, cache):
    """gcn3d.py:108-112 / 182-186 ORL_forward: conv2([f, g]) + f."""
    g = orl_global(feature, xyz, k, cache, name).repeat(1, feature.shape[1], 1)
    return _pointwise(torch.cat([feature, g], dim=-1), P[name + ".conv2.weight"]) + feature


def surface_conv(P, name, xyz, k, cache):
    """gcn3d.py:60-112 HSlayer_surface.forward."""
    S = P["_support_num"]
    f_ste = _pointwise(xyz, P[name + ".STE_layer.weight"])
    idx = cache.knn(name + ".rf", xyz, k)
    dirs = neighbor_directions(xyz, idx)                              # (B,n,k,3)
    sdn = F.normalize(P[name + ".directions"], dim=0)                # (3, S*C)
    theta = torch.relu(dirs @ sdn)
    B, n = xyz.shape[:2]
    theta = theta.reshape(B, n, k, S, -1).max(dim=2)[0].mean(dim=2)  # max over k, mean over S
    theta = tapv(P, name + ".g", theta)
    return tapv(P, name + ".out", _orl_forward(P, name, theta, xyz, k, cache) + f_ste)


def hs_conv(P, name, xyz, fmap, k, cache):
    """gcn3d.py:115-186 HS_layer.forward."""
    S = P["_support_num"]
    W, bias = P[name + ".weights"], P[name + ".bias"]
    cout = W.shape[1] // (S + 1)
    f_ste = _pointwise(fmap, P[name + ".STE_layer.weight"])
    idx = cache.knn(name + ".rf", fmap, k)                            # feature-space graph (RF-F)
    dirs = neighbor_directions(xyz, idx)
    sdn = F.normalize(P[name + ".directions"], dim=0)
    B, n = xyz.shape[:2]
    theta = torch.relu(dirs @ sdn).reshape(B, n, k, -1)               # (B,n,k,S*cout)
    proj = tapv(P, name + ".proj", fmap @ W + bias)                   # (B,n,(S+1)*cout)
    center, support = proj[:, :, :cout], proj[:, :, cout:]
    act = (theta * gather_rows(support, idx)).view(B, n, k, S, cout)
    feature = tapv(P, name + ".g", center + act.max(dim=2)[0].mean(dim=2))
    return tapv(P, name + ".out", _orl_forward(P, name, feature, xyz, k, cache) + f_ste)


def pool(xyz, fmap, sample_idx, cache, tag, k=4):
    """gcn3d.py:219-245 Pool_layer.forward with the random subsample passed in explicitly."""
    idx = cache.knn(tag + ".xyz", xyz, k)
    pooled = gather_rows(fmap, idx).max(dim=2)[0]
    return xyz[:, sample_idx, :], pooled[:, sample_idx, :]


def draw_sample_idx(n, rate=4):
    """gcn3d.py:241-242: ``torch.randperm(n)[:int(n / rate)]`` from the global CPU generator."""
    return torch.randperm(n)[: int(n / rate)]
