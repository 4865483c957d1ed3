"""Drop-in for ``network/fs_net_repo/PoseR.py``: rotation-axis heads on the HIP kernels.

Rot_green (y axis) / Rot_red (x axis): Conv1d 1286->1024->256 (+BN, ReLU), max over points,
256->256 (+BN, ReLU), dropout (eval: identity), 256->4 = confidence + axis.  Reference PoseR.py:10-70.
"""
import torch
import torch.nn as nn

from ... import engine
from ...config import FLAGS
from .gcn3d import _Packable


class _PointHead(_Packable):
    """Parameter container with the reference's names (conv1..4, bn1..3, drop1); fused eval forward, differentiable
    training forward (tgpose_amd.autograd.point_head: batch-statistics BatchNorm, dropout, HIP backward)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.f, self.k = in_channels, out_channels
        self.conv1 = nn.Conv1d(self.f, 1024, 1)
        self.conv2 = nn.Conv1d(1024, 256, 1)
        self.conv3 = nn.Conv1d(256, 256, 1)
        self.conv4 = nn.Conv1d(256, self.k, 1)
        self.drop1 = nn.Dropout(0.2)
        self.bn1 = nn.BatchNorm1d(1024)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(256)

    def _sig(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _run(self, x):
        """x: (B, C, N) as in the reference -> (B, k).  C may be 1286 or 1289."""
        B, C, N = x.shape
        if self.training:
            from ... import autograd as tgp_autograd
            return tgp_autograd.point_head(self, torch.nn.functional.pad(x.float().transpose(1, 2), (0, engine.FEAT_LD - C)))
        pk = self._packed(lambda: engine.pack_head(engine._dev_sd(self.state_dict(), x.device), ""))
        rows = torch.zeros(B, N, engine.FEAT_LD, device=x.device, dtype=torch.float32)
        rows[:, :, :C].copy_(x.detach().float().transpose(1, 2))
        return engine.head_forward(pk, rows, N)


class Rot_green(_PointHead):
    def __init__(self):
        super().__init__(FLAGS.feat_c_R, FLAGS.R_c)

    def forward(self, x):
        return self._run(x)


class Rot_red(_PointHead):
    def __init__(self):
        super().__init__(FLAGS.feat_c_R, FLAGS.R_c)

    def forward(self, x):
        return self._run(x)
