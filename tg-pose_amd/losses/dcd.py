"""Drop-in for the Chamfer-based pieces of ``losses/TDA_loss_sym_recon.py`` on the HIP kernels:
``calc_cd`` (:495-509), ``calc_dcd`` (:411-450) and ``TDA_loss.R_DCD`` (:326-342), all differentiable.

``calc_dcd`` is differentiable w.r.t. the predicted cloud exactly as the reference is: the density weights are
detached, the gradient flows through exp(-alpha d) into the Chamfer backward (``tgp_dcd_bwd`` ->
``tgp_chamfer_bwd``).  The reference's Python loop over the batch with ``torch.bincount`` per sample becomes one
launch with one workgroup per object.
"""
import math

import torch
from torch.autograd import Function

from .. import ops
from .chamfer3D.dist_chamfer_3D import chamfer_3DDist


def calc_cd(pred, gt, return_raw=False, separate=False):
    dist1, dist2, idx1, idx2 = chamfer_3DDist()(pred, gt)
    if separate:
        res = [torch.cat([torch.sqrt(dist1).mean(1).unsqueeze(0), torch.sqrt(dist2).mean(1).unsqueeze(0)]),
               torch.cat([dist1.mean(1).unsqueeze(0), dist2.mean(1).unsqueeze(0)])]
    else:
        res = [(torch.sqrt(dist1).mean(1) + torch.sqrt(dist2).mean(1)) / 2, dist1.mean(1) + dist2.mean(1)]
    if return_raw:
        res.extend([dist1, dist2, idx1, idx2])
    return res


class _DcdFunction(Function):
    @staticmethod
    def forward(ctx, pred, gt, alpha, n_lambda, non_reg):
        B, n, _ = pred.shape
        m = gt.shape[1]
        dev = pred.device
        dist1, dist2 = torch.zeros(B, n, device=dev), torch.zeros(B, m, device=dev)
        idx1 = torch.zeros(B, n, device=dev, dtype=torch.int32)
        idx2 = torch.zeros(B, m, device=dev, dtype=torch.int32)
        ops.chamfer_fwd(pred, gt, dist1, dist2, idx1, idx2)
        loss, w1, w2 = ops.dcd_fwd(dist1, dist2, idx1, idx2, alpha, n_lambda, non_reg)
        ctx.save_for_backward(pred, gt, dist1, dist2, idx1, idx2, w1, w2)
        ctx.alpha = alpha
        ctx.mark_non_differentiable(idx1, idx2, w1, w2)
        return loss, dist1, dist2, idx1, idx2, w1, w2

    @staticmethod
    def backward(ctx, gloss, gdist1, gdist2, gi1, gi2, gw1, gw2):
        pred, gt, dist1, dist2, idx1, idx2, w1, w2 = ctx.saved_tensors
        gd1, gd2 = ops.dcd_bwd(dist1, dist2, w1, w2, gloss.contiguous().float(), ctx.alpha)
        if gdist1 is not None:
            gd1 = gd1 + gdist1
        if gdist2 is not None:
            gd2 = gd2 + gdist2
        g_pred, g_gt = torch.zeros_like(pred), torch.zeros_like(gt)
        ops.chamfer_bwd(pred, gt, g_pred, g_gt, gd1.contiguous(), gd2.contiguous(), idx1, idx2)
        return g_pred, g_gt, None, None, None


def calc_dcd(pred_recon, cate_gt, alpha=0.1, n_lambda=0.3, return_raw=False, non_reg=False):
    """-> per-object loss (B,) [+ dist1, dist2, idx1, idx2 when return_raw]"""
    pred = pred_recon.float().contiguous()
    gt = cate_gt.float().contiguous()
    assert pred.shape[0] == gt.shape[0]
    loss, dist1, dist2, idx1, idx2, _, _ = _DcdFunction.apply(pred, gt, float(alpha), float(n_lambda), bool(non_reg))
    if return_raw:
        return [loss, dist1, dist2, idx1, idx2]
    return loss


def recon_completion(pred_recon, cate_gt, alpha=0.1, n_lambda=0.3, return_raw=False, non_reg=False):
    """losses/TDA_loss_sym_recon.py:453-490: calc_dcd's two directed terms combined as 0.9 mean_b(loss1) + 0.1 mean_b(loss2), a
    scalar.  The Chamfer search, the density weights (detached, as there) and the backward through exp(-alpha d) are the DCD
    kernels'; the two means over (B, n) / (B, m) are element-wise torch ops on their outputs (the term is in none of the name
    lists the reference's trainer builds, engine/organize_loss.py: kept for completeness of the seam, not tuned)."""
    pred = pred_recon.float().contiguous()
    gt = cate_gt.float().contiguous()
    assert pred.shape[0] == gt.shape[0]
    _, dist1, dist2, idx1, idx2, w1, w2 = _DcdFunction.apply(pred, gt, float(alpha), float(n_lambda), bool(non_reg))
    loss1 = (-torch.exp(-dist1 * alpha) * w1 + 1.).mean(1)
    loss2 = (-torch.exp(-dist2 * alpha) * w2 + 1.).mean(1)
    res = 0.9 * torch.mean(loss1) + 0.1 * torch.mean(loss2)
    if return_raw:
        return [res, dist1, dist2, idx1, idx2]
    return res


class _PoseTransform(Function):
    """(R^T (points - t)) * s with gradients for all four inputs (HIP forward and backward)"""

    @staticmethod
    def forward(ctx, points, R, t, s):
        ctx.save_for_backward(points, R, t, s)
        return ops.pose_transform(points, R, t, s)

    @staticmethod
    def backward(ctx, dout):
        points, R, t, s = ctx.saved_tensors
        dp, dR, dt, ds = ops.pose_transform_bwd(points, R, t, s, dout, need_points=ctx.needs_input_grad[0])
        return dp, dR, dt, ds


def _rodrigues(k, angle, v):
    """v rotated about the unit axis k by `angle` (the matrix of to_rot_matrix_in_batch applied to v)"""
    c, s_ = torch.cos(angle), torch.sin(angle)
    return v * c + torch.cross(k, v, dim=-1) * s_ + k * (k * v).sum(-1, keepdim=True) * (1 - c)


def _vertical_axes(c1, c2, y, z):
    """get_vertical_rot_vec_in_batch (TDA_loss_sym_recon.py:370-395) on (B,3) tensors: rotate y and z about their common
    normal until they are perpendicular, sharing the correction in proportion to the other axis' confidence"""
    c1, c2 = c1.unsqueeze(-1), c2.unsqueeze(-1)
    k = torch.cross(y, z, dim=-1)
    k = k / (torch.norm(k, dim=-1, keepdim=True) + 1e-8)
    theta = torch.acos(torch.clamp((y * z).sum(-1, keepdim=True), -1 + 1e-6, 1 - 1e-6)) - math.pi / 2
    return _rodrigues(k, c2 / (c1 + c2) * theta, y), _rodrigues(k, -(c1 / (c1 + c2)) * theta, z)


class _PoseRotation(Function):
    """pose_rotation_torch as one launch forward (R and its Jacobian by forward-mode differentiation in the kernel) and one backward
    (csrc/poserot.hip): ~300 element-wise launches less per training step"""

    @staticmethod
    def forward(ctx, gR0, p_g, f_g, p_r, f_r, sym0):
        R, J = ops.pose_rotation_fwd(gR0, p_g, f_g, p_r, f_r, sym0)
        ctx.save_for_backward(J)
        return R

    @staticmethod
    def backward(ctx, dR):
        (J,) = ctx.saved_tensors
        din = ops.pose_rotation_bwd(dR.contiguous(), J)
        return None, din[:, 0:3], din[:, 3], din[:, 4:7], din[:, 7], None


def pose_rotation(g_R, p_g_vec, f_g_vec, p_r_vec, f_r_vec, sym):
    """The rotation R_DCD canonicalises with (TDA_loss_sym_recon.py:327-333, get_rot_mat_y_first :351-360): (B,3,3),
    differentiable w.r.t. the predicted axes and confidences."""
    c = lambda t: t.float().contiguous()
    return _PoseRotation.apply(c(g_R[..., 0]), c(p_g_vec), c(f_g_vec).reshape(-1), c(p_r_vec), c(f_r_vec).reshape(-1),
                               c(sym[:, 0]))


def pose_rotation_torch(g_R, p_g_vec, f_g_vec, p_r_vec, f_r_vec, sym):
    """pose_rotation as (B,3)-sized torch arithmetic under autograd: the formulation the fused kernel is tested against
    (tests/test_gpu_parity.py::test_pose_rotation_fused_vs_torch_autograd)"""
    ys, xs = _vertical_axes(f_g_vec, torch.full_like(f_g_vec, 1e-5), p_g_vec, g_R[..., 0])
    y, x = _vertical_axes(f_g_vec, f_r_vec, p_g_vec, p_r_vec)
    flag = sym[:, 0].unsqueeze(-1) == 1
    y, x = torch.where(flag, ys, y), torch.where(flag, xs, x)
    y = torch.nn.functional.normalize(y, dim=-1)
    z = torch.nn.functional.normalize(torch.cross(x, y, dim=-1), dim=-1)
    return torch.stack((torch.cross(y, z, dim=-1), y, z), dim=-1)


def R_DCD(cate_ori, points, g_R, p_g_vec, f_g_vec, p_r_vec, f_r_vec, p_t, p_s, sym):
    """TDA_loss.R_DCD (:326-342): canonicalise the reconstruction with the predicted pose, density-aware Chamfer against the
    category prior (alpha 70, lambda 0.3), mean over the batch.  Differentiable w.r.t. the reconstruction and every pose
    input, as in the reference."""
    R = pose_rotation(g_R.float(), p_g_vec.float(), f_g_vec.float(), p_r_vec.float(), f_r_vec.float(), sym.float())
    canon = _PoseTransform.apply(points.float(), R, p_t.float(), p_s.float())
    return torch.mean(calc_dcd(canon, cate_ori, alpha=70, n_lambda=0.3, return_raw=False, non_reg=False))
