"""Drop-in for the mAP computation of ``evaluation/eval_utils_v1.py`` (compute_degree_cm_mAP :1227-1548 and what it calls).

The reference evaluates image by image and class by class in Python: for every (prediction, ground truth) pair a 3D-IoU
(20 box rotations for the symmetric categories) and a rotation / translation error, then a greedy matching that is
repeated for each of the 101 IoU and 62 x 22 pose thresholds.  Here

  1. one host pass collects every pair of the whole result set,
  2. two HIP launches (``tgp_iou3d_pairs``, ``tgp_rt_error_pairs``; double precision, as numpy) compute all pair metrics,
  3. the greedy matchings run vectorised over the threshold axes (numpy; the loops that remain are over the handful of
     instances of one image and class),
  4. the AP integration is the reference's formula.

Same inputs (the ``final_results`` list the evaluater pickles, evaluation/evaluate.py:53-67), same outputs
(``iou_3d_aps`` (classes+1, iou thresholds), ``pose_aps`` (classes+1, degree thresholds+1, shift thresholds+1)) and the same
``mAP_data.npz`` in ``log_dir``.  The figure the reference also draws is not produced.
"""
import os

import numpy as np
import torch

from .. import _lib
from ..ops import _p, _stream, check

_SYM_IOU = ("bottle", "bowl", "can")          # compute_3d_iou_new :864
_SYM_ROT = ("bottle", "can", "bowl")          # compute_RT_degree_cm_symmetry :935
_HALF_TURN = ("phone", "eggbox", "glue")      # :949


def pair_metrics(RT1, RT2, scales1, scales2, symmetric, rot_mode, device="cuda"):
    """All pairs at once.  RT1, RT2 (P,4,4), scales (P,3), symmetric (P,) {0,1}, rot_mode (P,) {0,1,2} (numpy) ->
    iou (P,) float64, degree_cm (P,2) float64."""
    P = len(RT1)
    if P == 0:
        return np.zeros(0), np.zeros((0, 2))
    dev = torch.device(device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)
    r1, r2 = t(RT1, torch.float64).reshape(P, 16), t(RT2, torch.float64).reshape(P, 16)
    s1, s2 = t(scales1, torch.float64).reshape(P, 3), t(scales2, torch.float64).reshape(P, 3)
    sym, mode = t(symmetric, torch.int32), t(rot_mode, torch.int32)
    iou = torch.empty(P, dtype=torch.float64, device=dev)
    err = torch.empty(P, 2, dtype=torch.float64, device=dev)
    check(_lib.lib().tgp_iou3d_pairs(_p(r1), _p(r2), _p(s1), _p(s2), _p(sym), P, _p(iou), _stream(r1)), "tgp_iou3d_pairs")
    check(_lib.lib().tgp_rt_error_pairs(_p(r1), _p(r2), _p(mode), P, _p(err), _stream(r1)), "tgp_rt_error_pairs")
    return iou.cpu().numpy(), err.cpu().numpy()


def _iou_matches(overlaps, thresholds):
    """compute_3d_matches' greedy loop (:1091-1122; every pair is of one class here), all thresholds at once.
    overlaps (P,G) float32, predictions already sorted by score -> gt_matches (T,G), pred_matches (T,P)."""
    P, G = overlaps.shape
    T = len(thresholds)
    # the reference compares a float32 array element with a Python float: under numpy >= 2 (NEP 50) that comparison is
    # made in float32, i.e. against the threshold rounded to float32
    thr = np.asarray(thresholds, dtype=np.float32)
    pred_matches = -1 * np.ones([T, P])
    gt_matches = -1 * np.ones([T, G])
    for i in range(P):
        order = np.argsort(overlaps[i])[::-1]
        low = np.where(overlaps[i, order] < 0)[0]                    # score_threshold = 0
        if low.size > 0:
            order = order[:low[0]]
        active = np.ones(T, dtype=bool)                              # thresholds whose inner loop has not ended yet
        for j in order:
            iou = overlaps[i, j]
            free = gt_matches[:, j] <= -1
            active &= ~(free & (iou < thr))                          # "break": the sorted IoUs only get smaller
            hit = active & free & (iou > thr)
            gt_matches[hit, j] = i
            pred_matches[hit, i] = j
            active &= ~hit
            if not active.any():
                break
    return gt_matches, pred_matches


def _pose_matches(err, degree_thres, shift_thres):
    """compute_match_from_degree_cm (:1182-1224; one class), all (degree, shift) thresholds at once.
    err (P,G,2) -> gt_matches (D,S,G), pred_matches (D,S,P)."""
    P, G = err.shape[:2]
    D, S = len(degree_thres), len(shift_thres)
    pred_matches = -1 * np.ones((D, S, P))
    gt_matches = -1 * np.ones((D, S, G))
    if P == 0 or G == 0:
        return gt_matches, pred_matches
    dt = np.asarray(degree_thres, dtype=np.float64)[:, None]
    st = np.asarray(shift_thres, dtype=np.float64)[None, :]
    for i in range(P):
        order = np.argsort(np.sum(err[i, :, :], axis=-1))
        todo = np.ones((D, S), dtype=bool)
        for j in order:
            ok = todo & (gt_matches[:, :, j] <= -1) & ~((err[i, j, 0] > dt) | (err[i, j, 1] > st))
            gt_matches[ok, j] = i
            pred_matches[ok, i] = j
            todo &= ~ok
            if not todo.any():
                break
    return gt_matches, pred_matches


def _ap(pred_match, pred_scores, gt_match):
    """compute_ap_from_matches_scores (:1127-1153)"""
    assert pred_match.shape[0] == pred_scores.shape[0]
    order = np.argsort(pred_scores)[::-1]
    pred_match = pred_match[order]
    hits = np.cumsum(pred_match > -1)
    precisions = hits / (np.arange(len(pred_match)) + 1)
    recalls = hits.astype(np.float32) / len(gt_match)
    precisions = np.concatenate([[0], precisions, [0]])
    recalls = np.concatenate([[0], recalls, [1]])
    precisions = np.maximum.accumulate(precisions[::-1])[::-1]       # the reference's backward running maximum
    idx = np.where(recalls[:-1] != recalls[1:])[0] + 1
    return np.sum((recalls[idx] - recalls[idx - 1]) * precisions[idx])


def compute_degree_cm_mAP(final_results, synset_names, log_dir=None, degree_thresholds=[360], shift_thresholds=[100],
                          iou_3d_thresholds=[0.1], iou_pose_thres=0.1, use_matches_for_pose=False, eval_recon=False,
                          plot_figure=False, device="cuda"):
    if plot_figure:
        raise NotImplementedError("the AP-curve figure of the reference is not drawn; pass plot_figure=False")
    if eval_recon:
        raise NotImplementedError("eval_recon (EMD / Chamfer statistics of the results) is not part of this path")
    num_classes = len(synset_names)
    degree_thres_list = list(degree_thresholds) + [360]
    shift_thres_list = list(shift_thresholds) + [100]
    iou_thres_list = list(iou_3d_thresholds)
    nD, nS, nI = len(degree_thres_list), len(shift_thres_list), len(iou_thres_list)
    if use_matches_for_pose:
        assert iou_pose_thres in iou_thres_list

    # ---- pass 1: every (image, class) group and its pairs
    groups, off = [], 0
    RT1, RT2, S1, S2, SYM, MODE = [], [], [], [], [], []
    for result in final_results:
        gt_class_ids = result['gt_class_ids'].astype(np.int32)
        gt_RTs, gt_scales = np.array(result['gt_RTs']), np.array(result['gt_scales'])
        gt_hv = result['gt_handle_visibility']
        pred_bboxes = np.array(result['pred_bboxes'])
        pred_class_ids, pred_scales = result['pred_class_ids'], result['pred_scales']
        pred_scores, pred_RTs = result['pred_scores'], np.array(result['pred_RTs'])
        if len(gt_class_ids) == 0 and len(pred_class_ids) == 0:
            continue
        for cls_id in range(1, num_classes):
            name = synset_names[cls_id]
            gsel = gt_class_ids == cls_id if len(gt_class_ids) else np.zeros(0, bool)
            psel = pred_class_ids == cls_id if len(pred_class_ids) else np.zeros(0, bool)
            g_RT = gt_RTs[gsel] if len(gt_class_ids) else np.zeros((0, 4, 4))
            g_sc = gt_scales[gsel] if len(gt_class_ids) else np.zeros((0, 3))
            p_box = pred_bboxes[psel, :] if len(pred_class_ids) else np.zeros((0, 4))
            p_sco = pred_scores[psel] if len(pred_class_ids) else np.zeros(0)
            p_RT = pred_RTs[psel] if len(pred_class_ids) else np.zeros((0, 4, 4))
            p_sc = pred_scales[psel] if len(pred_class_ids) else np.zeros((0, 3))
            G, P = len(g_RT), len(p_RT)
            if name != 'mug':
                hv = np.ones(G, dtype=np.int32)
            else:
                hv = np.asarray(gt_hv)[gsel] if len(gt_class_ids) else np.ones(0)
            if P:
                assert not np.all(p_box == 0, axis=1).any(), "zero-padded prediction boxes (trim_zeros asserts the same)"
                order = np.argsort(p_sco)[::-1]                          # compute_3d_matches :1069
                p_sco, p_RT, p_sc = p_sco[order], p_RT[order], p_sc[order]
            if P and G:
                sym = np.array([(name in _SYM_IOU) or (name == 'mug' and hv[j] == 0) for j in range(G)], dtype=np.int32)
                mode = np.array([1 if (name in _SYM_ROT or (name == 'mug' and hv[j] == 0)) else (2 if name in _HALF_TURN else 0)
                                 for j in range(G)], dtype=np.int32)
                RT1.append(np.repeat(p_RT, G, axis=0)), RT2.append(np.tile(g_RT, (P, 1, 1)))
                S1.append(np.repeat(p_sc, G, axis=0)), S2.append(np.tile(g_sc, (P, 1)))
                SYM.append(np.tile(sym, P)), MODE.append(np.tile(mode, P))
            groups.append((cls_id, P, G, off, p_sco))
            off += P * G

    # ---- pass 2: all pair metrics on the device
    cat = lambda xs, shape: np.concatenate(xs) if xs else np.zeros(shape)
    iou_all, err_all = pair_metrics(cat(RT1, (0, 4, 4)), cat(RT2, (0, 4, 4)), cat(S1, (0, 3)), cat(S2, (0, 3)),
                                    cat(SYM, (0,)), cat(MODE, (0,)), device)
    iou_all = iou_all.astype(np.float32)                                 # the reference stores overlaps as float32 (:1078)

    # ---- pass 3: matching per group, accumulation per class
    iou_pm = [[] for _ in range(num_classes)]
    iou_ps = [[] for _ in range(num_classes)]
    iou_gm = [[] for _ in range(num_classes)]
    pose_pm = [[] for _ in range(num_classes)]
    pose_ps = [[] for _ in range(num_classes)]
    pose_gm = [[] for _ in range(num_classes)]
    for cls_id, P, G, o, p_sco in groups:
        overlaps = iou_all[o:o + P * G].reshape(P, G)
        err = err_all[o:o + P * G].reshape(P, G, 2)
        gm, pm = _iou_matches(overlaps, iou_thres_list)
        iou_pm[cls_id].append(pm), iou_gm[cls_id].append(gm)
        iou_ps[cls_id].append(np.tile(p_sco, (nI, 1)))
        keep_p, keep_g = np.ones(P, bool), np.ones(G, bool)
        if use_matches_for_pose:
            ti = iou_thres_list.index(iou_pose_thres)
            keep_p, keep_g = pm[ti] > -1, gm[ti] > -1
        gmp, pmp = _pose_matches(err[keep_p][:, keep_g], degree_thres_list, shift_thres_list)
        pose_pm[cls_id].append(pmp), pose_gm[cls_id].append(gmp)
        pose_ps[cls_id].append(np.tile(p_sco[keep_p], (nD, nS, 1)))

    iou_3d_aps = np.zeros((num_classes + 1, nI))
    pose_aps = np.zeros((num_classes + 1, nD, nS))
    for cls_id in range(1, num_classes):
        pm = np.concatenate(iou_pm[cls_id] or [np.zeros((nI, 0))], axis=-1)
        ps = np.concatenate(iou_ps[cls_id] or [np.zeros((nI, 0))], axis=-1)
        gm = np.concatenate(iou_gm[cls_id] or [np.zeros((nI, 0))], axis=-1)
        for s in range(nI):
            iou_3d_aps[cls_id, s] = _ap(pm[s], ps[s], gm[s])
        pm = np.concatenate(pose_pm[cls_id] or [np.zeros((nD, nS, 0))], axis=-1)
        ps = np.concatenate(pose_ps[cls_id] or [np.zeros((nD, nS, 0))], axis=-1)
        gm = np.concatenate(pose_gm[cls_id] or [np.zeros((nD, nS, 0))], axis=-1)
        for i in range(nD):
            for j in range(nS):
                pose_aps[cls_id, i, j] = _ap(pm[i, j], ps[i, j], gm[i, j])
    iou_3d_aps[-1, :] = np.mean(iou_3d_aps[1:-1, :], axis=0)
    pose_aps[-1] = np.mean(pose_aps[1:-1], axis=0)
    if log_dir is not None:
        np.savez(os.path.join(log_dir, 'mAP_data.npz'), pose_aps=pose_aps, degree_thres_list=degree_thres_list,
                 shift_thres_list=shift_thres_list, iou_thres_list=iou_thres_list, iou_3d_aps=iou_3d_aps)
    return iou_3d_aps, pose_aps
