"""Oracle (TEST INFRASTRUCTURE): the regression terms of the training loss restated in plain CPU torch, object by object as the
reference's loops run -- TDA_loss's small terms (losses/TDA_loss_sym_recon.py:205-290), the persistence-image terms
(:292-322), prop_sym_matching_loss and feat_consistency_loss (losses/consistency_loss.py:11-81).  Everything is ordinary
differentiable torch, so autograd on this file supplies the reference gradients.  Pinned by tests/golden/tda_loss.npz (values
and gradients produced by the imported reference)."""
import math

import torch
import torch.nn.functional as F

WEIGHTS = dict(rot_1_w=8.0, rot_2_w=8.0, rot_regular=4.0, tran_w=8.0, size_w=8.0, r_con_w=1.0, h1_w=4.0, h2_w=4.0, feat_consist_w=2.0,
               DCD_align=1.0, prop_sym_w=1.0)                                       # config/config.py:71-114


def penalty(kind, beta=0.5):
    """nn.L1Loss / nn.SmoothL1Loss(beta) with mean reduction (:20-35)"""
    if kind == "l1":
        return lambda a, b: (a - b).abs().mean()
    return lambda a, b: F.smooth_l1_loss(a, b, beta=beta)


def pose_terms(pred, gt, sym, kind="l1"):
    """-> dict of the eight unweighted terms.  pred: Rot1, Rot2 (B,3), Rot1_f, Rot2_f (B), Tran, Size (B,3); gt: Rot1, Rot2, Tran, Size"""
    rho = penalty(kind)
    B = pred["Rot1"].shape[0]
    out = {"Rot1": rho(pred["Rot1"], gt["Rot1"]),                                             # :223-225
           "Rot1_cos": ((1 - (pred["Rot1"] * gt["Rot1"]).sum(1)) * 2).mean(),                 # :243-245
           "Tran": rho(pred["Tran"], gt["Tran"]), "Size": rho(pred["Size"], gt["Size"])}     # :285-290
    rot2 = cos2 = reg = torch.zeros(())
    valid = 0
    for i in range(B):                                                                        # :227-282: objects symmetric about y are skipped
        if int(sym[i, 0]) == 1:
            continue
        rot2 = rot2 + rho(pred["Rot2"][i], gt["Rot2"][i])
        cos2 = cos2 + (1.0 - (pred["Rot2"][i] * gt["Rot2"][i]).sum()) * 2.0
        reg = reg + torch.dot(pred["Rot1"][i], pred["Rot2"][i]).abs()
        valid += 1
    d = max(valid, 1)
    out["Rot2"], out["Rot2_cos"], out["Rot_regular"] = rot2 / d, cos2 / d, reg / d
    ng = (pred["Rot1"] - gt["Rot1"]).norm(dim=-1)                                             # :205-221
    res_g = rho(torch.exp(-13.7 * ng * ng), pred["Rot1_f"])
    res_r = torch.zeros(())
    for i in range(B):
        if int(sym[i, 0]) == 0:
            nr = (pred["Rot2"][i] - gt["Rot2"][i]).norm()
            res_r = res_r + rho(torch.exp(-13.7 * nr * nr), pred["Rot2_f"][i])
    out["R_con"] = res_g + res_r / B
    return out


def prop_sym_matching_loss(PC, PC_re, gt_R, gt_t, sym):
    """consistency_loss.py:19-81: per object the cloud is taken to the ground-truth object frame, flipped by the object's symmetry
    and posed back; objects symmetric about y with no further flag contribute zero on both sides"""
    B = PC.shape[0]
    tgt, rec = [], []
    for b in range(B):
        s0, rest, s1 = int(sym[b, 0]), int(sym[b, 1:].sum()), int(sym[b, 1])
        cano = (PC[b] - gt_t[b]) @ gt_R[b]                                                    # rows = R^T (p - t)
        zero = torch.zeros_like(PC[b])
        if s0 == 1 and rest > 0:
            t = (cano * torch.tensor([-1.0, 1.0, -1.0])) @ gt_R[b].T + gt_t[b]
        elif s0 == 0 and s1 == 1:
            t = (cano * torch.tensor([1.0, 1.0, -1.0])) @ gt_R[b].T + gt_t[b]
        elif s0 == 0:
            t = PC[b]
        else:
            t = zero
        tgt.append(t)
        rec.append(zero if (s0 == 1 and rest == 0) else PC_re[b])
    return (torch.stack(tgt) - torch.stack(rec)).abs().mean()


def _bad(t):
    return bool(torch.isnan(t).any() or torch.isinf(t).any())


def ph_loss(ph, gt_ph):
    """ph_loss_fn (:292-298)"""
    w = (gt_ph.sum(1, keepdim=True) > 0) * 1.0
    if _bad(ph) or _bad(gt_ph):
        return ((ph - ph).abs() * w).mean()
    return ((ph - gt_ph).abs() * w).mean()


def omega(gt_h, cate_h, k=2, lam=1):
    """:313-322 -- a Python float in the reference"""
    return k * math.exp(-lam * float(ph_loss(gt_h.detach(), cate_h.detach())))


def ph_loss_cate(ph, gt_ph, cate_ph):
    """ph_loss_fn_cate (:301-311)"""
    return ph_loss(ph, gt_ph) * omega(gt_ph, cate_ph)


def feat_consistency(x1, x2):
    """consistency_loss.py:11-15 without the weight"""
    return 2 - 2 * (F.normalize(x1, dim=1) * F.normalize(x2, dim=1)).sum() / x1.shape[0]
