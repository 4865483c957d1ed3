"""Drop-in for the operator seam ``network/fs_net_repo/gcn3d.py`` of TG-Pose, on the HIP kernels.

Same names, arguments and return conventions as the reference (int64 indices, freshly allocated
channel-last tensors on the input's device); each call enqueues gfx950 kernels through the C ABI
(include/tgpose.h).  As in the reference the operators are autograd-tracked: when gradients are
being recorded and a feature input or a layer parameter requires them, the call runs as the
``torch.autograd.Function``s of ``tgpose_amd.autograd`` (HIP forward AND HIP backward) and returns a
graph-attached tensor; otherwise the fused no-autograd kernels of ``tgpose_amd.engine`` run.  The
layers hold no BatchNorm / dropout, so ``.train()`` / ``.eval()`` make no difference to them.
Gradients with respect to the point coordinates (``vertices``) are not produced -- the trainer's
clouds are data (trainer/RL_TDA.py:111-116) -- and a ``vertices`` that requires grad is refused.

Reference: get_neighbor_index :14, get_nearest_index :26, indexing_neighbor_new :38,
HSlayer_surface :60, HS_layer :115, get_ORL_global :210, Pool_layer :219.
"""
import math

import torch
import torch.nn as nn

from ... import engine, ops


def _tracked(*tensors):
    """does this call have to be recorded by autograd?"""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _xyz(vertices):
    if vertices.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("gradients with respect to the point coordinates are not implemented (the clouds are data)")
    return vertices.detach().float().contiguous()


class _SeamGraphs(object):
    """neighbour lists of one layer call, in the calling convention of tgpose_amd.autograd's layer functions:
    level None = feature-space kNN of x, otherwise the xyz kNN (computed once per call and shared, where the reference
    recomputes it: gcn3d.py:85/213)"""

    def __init__(self):
        self.xyz = None

    def __call__(self, name, level, x, k):
        with torch.no_grad():
            if level is None:
                return ops.knn_feat(x.detach().float().contiguous(), k)
            if self.xyz is None:
                self.xyz = ops.knn_xyz(x.detach().float().contiguous(), k)
            return self.xyz


def get_neighbor_index(vertices, neighbor_num):
    """(bs, v, d) -> (bs, v, neighbor_num) int64; d == 3 or a multiple of 32 (feature space).  Integer output: no gradient."""
    v = vertices.detach().float()
    idx = ops.knn_xyz(v.contiguous(), neighbor_num) if v.shape[2] == 3 else ops.knn_feat(v, neighbor_num)
    return idx.long()


def get_nearest_index(target, source):
    """(bs, v1, 3), (bs, v2, 3) -> (bs, v1, 1) int64"""
    return ops.nn1(target.detach().float(), source.detach().float()).long().unsqueeze(-1)


def indexing_neighbor_new(tensor, index):
    """(bs, v, C), (bs, m, k) -> (bs, m, k, C): row gather (C a multiple of 4); differentiable in `tensor`."""
    bs, v, C = tensor.shape
    _, m, k = index.shape
    flat = index.reshape(bs, m * k).to(torch.int32).contiguous()
    if _tracked(tensor):
        from ... import autograd as tgp_autograd
        return tgp_autograd._GatherRows.apply(tensor.float(), flat).view(bs, m, k, C)
    out = torch.empty(bs, m * k, C, device=tensor.device, dtype=torch.float32)
    ops.gather_rows(tensor.detach().float().contiguous(), flat, out)
    return out.view(bs, m, k, C)


def get_ORL_global(feature, vertices, neighbor_num):
    """(bs, v, C), (bs, v, 3) -> (bs, v, C): neighbour max, mean over points, repeated per point; differentiable in `feature`."""
    idx = ops.knn_xyz(_xyz(vertices), neighbor_num)
    if _tracked(feature):
        from ... import autograd as tgp_autograd
        g = tgp_autograd._NbrMaxMean.apply(feature.float(), idx)
    else:
        g = ops.orl_global(feature.detach().float().contiguous(), idx)
    return g.unsqueeze(1).repeat(1, feature.shape[1], 1)


class _Packable(nn.Module):
    """Caches the kernel-ready form of a layer's weights; rebuilt when a parameter changes."""

    def _sig(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _packed(self, build):
        sig = self._sig()
        if getattr(self, "_pk_sig", None) != sig:
            self._pk, self._pk_sig = build(), sig
        return self._pk


class HSlayer_surface(_Packable):
    def __init__(self, kernel_num, support_num):
        super().__init__()
        self.kernel_num, self.support_num = kernel_num, support_num
        self.directions = nn.Parameter(torch.empty(3, support_num * kernel_num))
        self.STE_layer = nn.Conv1d(3, kernel_num, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(2 * kernel_num, kernel_num, kernel_size=1, bias=False)
        stdv = 1.0 / math.sqrt(support_num * kernel_num)
        self.directions.data.uniform_(-stdv, stdv)

    def _build(self):
        C = self.kernel_num
        w2 = self.conv2.weight.detach()[:, :, 0]
        ste = engine._pad_cols(self.STE_layer.weight.detach()[:, :, 0], 4)
        # conv2's feature half and the STE convolution as one operand over [g | x y z 0] (engine.surface_layer)
        return dict(C=C, sdn=ops.normalize_dirs(self.directions.detach()), ste=ste,
                    w1=w2[:, :C].contiguous(), w2=w2[:, C:].contiguous(), w1x=torch.cat([w2[:, :C], ste], dim=1).contiguous())

    def forward(self, vertices, neighbor_num):
        xyz = _xyz(vertices)
        if _tracked(*self.parameters()):
            from ... import autograd as tgp_autograd
            return tgp_autograd._surface(self, xyz, _SeamGraphs(), neighbor_num)
        idx = ops.knn_xyz(xyz, neighbor_num)
        out = torch.empty(xyz.shape[0], xyz.shape[1], self.kernel_num, device=xyz.device, dtype=torch.float32)
        return engine.surface_layer(self._packed(self._build), xyz, idx, idx, out)


class HS_layer(_Packable):
    def __init__(self, in_channel, out_channel, support_num):
        super().__init__()
        self.in_channel, self.out_channel, self.support_num = in_channel, out_channel, support_num
        self.weights = nn.Parameter(torch.empty(in_channel, (support_num + 1) * out_channel))
        self.bias = nn.Parameter(torch.empty((support_num + 1) * out_channel))
        self.directions = nn.Parameter(torch.empty(3, support_num * out_channel))
        self.STE_layer = nn.Conv1d(in_channel, out_channel, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(2 * out_channel, out_channel, kernel_size=1, bias=False)
        stdv = 1.0 / math.sqrt(out_channel * (support_num + 1))
        for p in (self.weights, self.bias, self.directions):
            p.data.uniform_(-stdv, stdv)

    def _build(self):
        C = self.out_channel
        w2 = self.conv2.weight.detach()[:, :, 0]
        dev = self.weights.device
        return dict(C=C, Cin=self.in_channel, sdn=ops.normalize_dirs(self.directions.detach()),
                    wcat=torch.cat([self.weights.detach().t(), self.STE_layer.weight.detach()[:, :, 0]], 0).contiguous(),
                    bcat=torch.cat([self.bias.detach(), torch.zeros(C, device=dev)]).contiguous(),
                    w1=w2[:, :C].contiguous(), w2=w2[:, C:].contiguous())

    def forward(self, vertices, feature_map, neighbor_num):
        xyz = _xyz(vertices)
        if _tracked(feature_map, *self.parameters()):
            from ... import autograd as tgp_autograd
            return tgp_autograd._hs(self, "conv", xyz, feature_map.float(), _SeamGraphs(), 0, neighbor_num)
        fmap = feature_map.detach().float().contiguous()
        out = torch.empty(xyz.shape[0], xyz.shape[1], self.out_channel, device=xyz.device, dtype=torch.float32)
        return engine.hs_layer(self._packed(self._build), xyz, fmap, ops.knn_feat(fmap, neighbor_num),
                               ops.knn_xyz(xyz, neighbor_num), out)


class Pool_layer(nn.Module):
    def __init__(self, pooling_rate=4, neighbor_num=4):
        super().__init__()
        self.pooling_rate, self.neighbor_num = pooling_rate, neighbor_num

    def forward(self, vertices, feature_map):
        xyz = _xyz(vertices)
        n = xyz.shape[1]
        idx = ops.knn_xyz(xyz, self.neighbor_num)
        sample = torch.randperm(n)[: int(n / self.pooling_rate)]      # global CPU generator, as gcn3d.py:242
        sample = sample.to(device=xyz.device, dtype=torch.int32)
        if _tracked(feature_map):
            from ... import autograd as tgp_autograd
            return tgp_autograd._PoolMax.apply(xyz, feature_map.float(), idx, sample, self.neighbor_num)
        return ops.pool(xyz, feature_map.detach().float().contiguous(), idx, sample, kpool=self.neighbor_num)
