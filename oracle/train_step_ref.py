"""Oracle (TEST INFRASTRUCTURE): the trainer's step -- ``RT_TDA_Trainer.RL_TDA_train_step`` (trainer/RL_TDA.py:110-200) and the
total of the loop body (:209-214) -- composed from the oracle's CPU restatements (posenet_ref, loss_ref, tda_loss_ref) in plain
differentiable torch: net1 with gradients, net2 (only_encoder) under no_grad on the augmented cloud, feat_consistency_loss +
2 x prop_sym_matching_loss, the fourteen control_loss('TDA') terms, total = 0.1 (con + recon_1 + recon_consistency) + 0.9 sum(TDA).
Dropout is off (p = 0), BatchNorm uses batch statistics.  Pinned by tests/golden/train_step_b4_n256.npz (losses, total and
gradients recorded from the reference's own RL_TDA_train_step; tests/test_oracle_golden.py)."""
import torch

from . import loss_ref as L
from . import posenet_ref as PR
from . import tda_loss_ref as T

TDA_NAMES = ['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con', 'TDA_h1', 'TDA_h2', 'TDA_h1_cate',
             'TDA_h2_cate', 'Prop_sym', 'R_DCD_cate_pred']                      # engine/organize_loss.py:16-18


def tda_terms(pred, gt, sym, names=TDA_NAMES, kind="l1"):
    """TDA_loss.forward (losses/TDA_loss_sym_recon.py:40-108) with the flag weights of config/config.py -> dict of weighted terms
    under the reference's keys ('Rot_regular' is reported as 'Rot_r_a')"""
    W = T.WEIGHTS
    t = T.pose_terms(pred, gt, sym, kind)
    out = {}
    for n in names:
        if n == "Rot1":
            out[n] = W["rot_1_w"] * t["Rot1"]
        elif n == "Rot1_cos":
            out[n] = W["rot_1_w"] * t["Rot1_cos"]
        elif n == "Rot2":
            out[n] = W["rot_2_w"] * t["Rot2"]
        elif n == "Rot2_cos":
            out[n] = W["rot_2_w"] * t["Rot2_cos"]
        elif n == "Rot_regular":
            out["Rot_r_a"] = W["rot_regular"] * t["Rot_regular"]
        elif n == "Prop_sym":
            out[n] = W["prop_sym_w"] * T.prop_sym_matching_loss(gt["Recon"], pred["Recon"], gt["R"], gt["Tran"], sym)
        elif n == "Tran":
            out[n] = W["tran_w"] * t["Tran"]
        elif n == "Size":
            out[n] = W["size_w"] * t["Size"]
        elif n == "R_con":
            out[n] = W["r_con_w"] * t["R_con"]
        elif n == "TDA_h1_cate":
            out[n] = T.ph_loss_cate(pred["TDA_h1"], gt["pdh1_category"], gt["h1"])
        elif n == "TDA_h1":
            out[n] = W["h1_w"] * T.ph_loss(pred["TDA_h1"], gt["h1"])
        elif n == "TDA_h2_cate":
            out[n] = T.ph_loss_cate(pred["TDA_h2"], gt["pdh2_category"], gt["h2"])
        elif n == "TDA_h2":
            out[n] = W["h2_w"] * T.ph_loss(pred["TDA_h2"], gt["h2"])
        elif n == "R_DCD_cate_pred":
            out[n] = W["DCD_align"] * L.r_dcd(gt["points_category"], pred["Recon"], gt["R"], pred["Rot1"], pred["Rot1_f"], pred["Rot2"],
                                               pred["Rot2_f"], pred["Tran"], pred["Size"], sym)
        else:
            raise KeyError(n)
    return out


def train_step(P1, P2, db, samples, inject=None, mode="exact"):
    """-> (loss_dict with 'RL_loss', 'recon_1_loss', 'recon_consistency_loss', 'TDA_loss' (dict), 'total'; results of net1; results of
    net2; graphs of both nets).  P1 / P2: reference-format state dicts (leaves that require grad collect the gradients of
    total.backward()); samples = [(pool_1, pool_2) of net1, (pool_1, pool_2) of net2]."""
    PC, obj = db["pcl_in"], db["cat_id"]
    r1, i1 = PR.posenet_forward(P1, PC, obj, sample_idx=samples[0], train_keys=True, mode=mode, inject=inject, bn_train=True,
                                want_intermediates=True)
    with torch.no_grad():
        r2, i2 = PR.encoder_only_forward(P2, db["aug_pcl_in"], obj, sample_idx=samples[1], mode=mode, inject=inject, bn_train=True,
                                         want_intermediates=True)
    gt_R, gt_t, sym = db["rotation"], db["translation"], db["sym_info"]
    ld = {"RL_loss": T.WEIGHTS["feat_consist_w"] * T.feat_consistency(r1["feat_global"], r2["feat_global"]),
          "recon_1_loss": T.prop_sym_matching_loss(PC, r1["recon"], gt_R, gt_t, sym),
          "recon_consistency_loss": 0.2 * T.prop_sym_matching_loss(r1["recon"], r2["recon"], gt_R, gt_t, sym)}
    pred = {"Rot1": r1["p_green_R"], "Rot1_f": r1["f_green_R"], "Rot2": r1["p_red_R"], "Rot2_f": r1["f_red_R"], "Recon": r1["recon"],
            "Tran": r1["Pred_T"], "Size": r1["Pred_s"], "TDA_h1": r1["h1"], "TDA_h2": r1["h2"]}
    gt = {"Rot1": gt_R[:, :, 1], "Rot2": gt_R[:, :, 0], "Recon": PC, "Tran": gt_t, "Size": db["fsnet_scale"], "h1": db["pdh1"],
          "h2": db["pdh2"], "pdh1_category": db["pdh1_category"], "pdh2_category": db["pdh2_category"],
          "points_category": db["points_category"], "R": gt_R}
    ld["TDA_loss"] = tda_terms(pred, gt, sym)
    ld["total"] = (0.1 * ld["RL_loss"] + 0.1 * ld["recon_1_loss"] + 0.1 * ld["recon_consistency_loss"]
                   + 0.9 * sum(v.sum() for v in ld["TDA_loss"].values()))
    graphs = dict(i1["indices"])
    graphs.update(i2["indices"])
    return ld, r1, r2, graphs
