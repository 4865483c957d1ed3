"""Drop-in for ``losses/chamfer3D/dist_chamfer_3D.py`` (native seam #3 of SURVEY.md section 8b).

``chamfer_3D`` below stands where the reference's pybind module of that name stands
(losses/chamfer3D/chamfer_cuda.cpp:30-33): ``forward(xyz1, xyz2, dist1, dist2, idx1, idx2)`` and
``backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)`` with the same
ownership rule -- the caller allocates every output, backward accumulates into the gradients --
and the same return value (1 on success).  ``chamfer_3DDist()(a, b) -> (dist1, dist2, idx1, idx2)``
is the Python face the loss uses (losses/TDA_loss_sym_recon.py:496-497).
"""
import torch
from torch import nn
from torch.autograd import Function

from ... import ops


class chamfer_3D(object):
    """Namespace with the two entry points of the reference's compiled extension."""

    @staticmethod
    def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
        return ops.chamfer_fwd(xyz1, xyz2, dist1, dist2, idx1, idx2)

    @staticmethod
    def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        return ops.chamfer_bwd(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)


class chamfer_3DFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        B, n, _ = xyz1.size()
        m = xyz2.size(1)
        dev = xyz1.device
        dist1 = torch.zeros(B, n, device=dev)            # allocated on the device directly: the
        dist2 = torch.zeros(B, m, device=dev)            # reference builds them on the CPU and copies
        idx1 = torch.zeros(B, n, device=dev, dtype=torch.int32)
        idx2 = torch.zeros(B, m, device=dev, dtype=torch.int32)
        chamfer_3D.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        chamfer_3D.backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1.contiguous(), graddist2.contiguous(), idx1, idx2)
        return gradxyz1, gradxyz2


class chamfer_3DDist(nn.Module):
    def forward(self, input1, input2):
        return chamfer_3DFunction.apply(input1.contiguous().float(), input2.contiguous().float())
