"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the Chamfer-based loss pieces of
``losses/TDA_loss_sym_recon.py`` -- calc_cd (:495-509), calc_dcd (:411-450), the axis/rotation helpers
(:351-408) and TDA_loss.R_DCD (:326-342) -- with the Chamfer search supplied by oracle/csrc/tgp_oracle.c.
Pinned by tests/golden/dcd.npz (values produced by the imported reference)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _clib


def chamfer(pred, gt):
    """Chamfer search by the C oracle.  When an input requires grad the distances are re-expressed through the found
    indices with torch ops (d1 = |pred - gt[idx1]|^2), which gives exactly the gradient the reference's Chamfer backward
    scatters (chamfer3D.cu:155-195) and lets torch autograd differentiate the losses built on top."""
    d1, d2, i1, i2 = _clib.chamfer_fwd(pred.detach().numpy(), gt.detach().numpy())
    i1, i2 = torch.from_numpy(i1.astype(np.int64)), torch.from_numpy(i2.astype(np.int64))
    if pred.requires_grad or gt.requires_grad:
        take = lambda src, idx: torch.gather(src, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
        return ((pred - take(gt, i1)) ** 2).sum(-1), ((gt - take(pred, i2)) ** 2).sum(-1), i1, i2
    return torch.from_numpy(d1), torch.from_numpy(d2), i1, i2


def calc_cd(pred, gt):
    d1, d2, _, _ = chamfer(pred, gt)
    return (d1.sqrt().mean(1) + d2.sqrt().mean(1)) / 2, d1.mean(1) + d2.mean(1)


def calc_dcd(pred, gt, alpha=70.0, n_lambda=0.3, non_reg=False):
    """per-object density-aware Chamfer loss (B,), plus the raw Chamfer outputs"""
    B, n, _ = pred.shape
    m = gt.shape[1]
    frac_12, frac_21 = n / m, m / n
    if non_reg:
        frac_12, frac_21 = max(1, frac_12), max(1, frac_21)
    d1, d2, i1, i2 = chamfer(pred, gt)
    e1, e2 = torch.exp(-d1 * alpha), torch.exp(-d2 * alpha)
    out = []
    for b in range(B):
        w1 = 1.0 / (torch.bincount(i1[b])[i1[b]].float() ** n_lambda + 1e-6) * frac_21
        w2 = 1.0 / (torch.bincount(i2[b])[i2[b]].float() ** n_lambda + 1e-6) * frac_12
        out.append((1.0 - e1[b] * w1).mean() + 0.5 * (1.0 - e2[b] * w2).mean())
    return torch.stack(out), (d1, d2, i1, i2)


def axis_angle_matrix(k, s, c):
    """Rodrigues matrices (B,3,3) about unit axes k (B,3) with sines/cosines s, c (B,1)   (:398-408)"""
    kx, ky, kz = k[:, 0:1], k[:, 1:2], k[:, 2:3]
    oc = 1 - c
    rows = [torch.cat([kx * kx * oc + c, kx * ky * oc - kz * s, kx * kz * oc + ky * s], -1),
            torch.cat([ky * kx * oc + kz * s, ky * ky * oc + c, ky * kz * oc - kx * s], -1),
            torch.cat([kx * kz * oc - ky * s, kz * ky * oc + kx * s, kz * kz * oc + c], -1)]
    return torch.stack(rows, dim=-2)


def vertical_axes(c1, c2, y, z):
    """get_vertical_rot_vec_in_batch (:370-395): make y, z perpendicular, sharing the correction by confidence"""
    c1, c2 = c1.unsqueeze(-1), c2.unsqueeze(-1)
    k = torch.cross(y, z, dim=-1)
    k = k / (torch.norm(k, dim=-1, keepdim=True) + 1e-8)
    theta = torch.acos(torch.clamp((y * z).sum(-1, keepdim=True), -1 + 1e-6, 1 - 1e-6))
    th2 = c1 / (c1 + c2) * (theta - math.pi / 2)
    th1 = c2 / (c1 + c2) * (theta - math.pi / 2)
    ny = torch.matmul(axis_angle_matrix(k, torch.sin(th1), torch.cos(th1)), y.unsqueeze(-1)).squeeze(-1)
    nz = torch.matmul(axis_angle_matrix(k, torch.sin(-th2), torch.cos(-th2)), z.unsqueeze(-1)).squeeze(-1)
    return ny, nz


def rot_from_y_x(y, x):
    """get_rot_mat_y_first (:351-360)"""
    y = F.normalize(y, dim=-1)
    z = F.normalize(torch.cross(x, y, dim=-1), dim=-1)
    return torch.stack((torch.cross(y, z, dim=-1), y, z), dim=-1)


def canonicalize(points, gR, p_g, f_g, p_r, f_r, p_t, p_s, sym):
    ys, xs = vertical_axes(f_g, torch.full_like(f_g, 1e-5), p_g, gR[..., 0])
    y, x = vertical_axes(f_g, f_r, p_g, p_r)
    flag = sym[:, 0].unsqueeze(-1) == 1
    R = rot_from_y_x(torch.where(flag, ys, y), torch.where(flag, xs, x))
    out = torch.matmul(R.transpose(-2, -1), (points - p_t.unsqueeze(-2)).transpose(-2, -1)).transpose(-2, -1)
    return out * p_s.unsqueeze(-2), R


def r_dcd(prior, points, gR, p_g, f_g, p_r, f_r, p_t, p_s, sym):
    canon, _ = canonicalize(points, gR, p_g, f_g, p_r, f_r, p_t, p_s, sym)
    return calc_dcd(canon, prior, 70.0, 0.3)[0].mean()
