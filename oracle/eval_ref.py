"""Oracle (TEST INFRASTRUCTURE): the two pairwise metrics of evaluation/eval_utils_v1.py restated pair by pair in numpy
float64 -- compute_3d_iou_new (:829-887) and compute_RT_degree_cm_symmetry (:890-963).  Pinned by tests/golden/eval_map.npz
(values produced by the imported reference)."""
import math

import numpy as np


def _corners(scale):
    sx, sy, sz = scale[0] / 2, scale[1] / 2, scale[2] / 2
    return np.array([[sx, sy, sz], [sx, sy, -sz], [-sx, sy, sz], [-sx, sy, -sz],
                     [sx, -sy, sz], [sx, -sy, -sz], [-sx, -sy, sz], [-sx, -sy, -sz]]).T      # (3, 8)


def _extent(RT, scale):
    """transform_coordinates_3d (:998-1012) then amax / amin over AXIS 0 of the [3, 8] corner array (:842-845): the
    reference reduces over the coordinate axis, so its "extent" is eight per-corner (min, max) pairs, not three per-axis
    ones.  Restated as written -- the AP numbers the reference reports come from exactly this."""
    h = RT @ np.vstack([_corners(scale), np.ones((1, 8))])
    pts = h[:3] / h[3]
    return pts.min(axis=0), pts.max(axis=0)


def _iou(RT1, RT2, s1, s2):
    lo1, hi1 = _extent(RT1, s1)
    lo2, hi2 = _extent(RT2, s2)
    d = np.minimum(hi1, hi2) - np.maximum(lo1, lo2)
    inter = 0.0 if d.min() < 0 else float(np.prod(d))
    return inter / (np.prod(hi1 - lo1) + np.prod(hi2 - lo2) - inter)


def iou_3d(RT1, RT2, s1, s2, symmetric):
    if not symmetric:
        return _iou(RT1, RT2, s1, s2)
    best = 0.0
    for i in range(20):
        th = 2 * math.pi * i / 20.0
        ry = np.array([[np.cos(th), 0, np.sin(th), 0], [0, 1, 0, 0], [-np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
        best = max(best, _iou(RT1 @ ry, RT2, s1, s2))
    return best


def rt_error(RT1, RT2, mode):
    """mode 0 general, 1 y-axis symmetric, 2 half-turn symmetric -> (degrees, cm)"""
    R1 = RT1[:3, :3] / np.cbrt(np.linalg.det(RT1[:3, :3]))
    R2 = RT2[:3, :3] / np.cbrt(np.linalg.det(RT2[:3, :3]))
    with np.errstate(invalid="ignore"):
        if mode == 1:
            y1, y2 = R1[:, 1], R2[:, 1]
            theta = np.arccos(y1.dot(y2) / (np.linalg.norm(y1) * np.linalg.norm(y2)))
        else:
            theta = np.arccos((np.trace(R1 @ R2.T) - 1) / 2)
            if mode == 2:
                theta = min(theta, np.arccos((np.trace(R1 @ np.diag([-1.0, 1.0, -1.0]) @ R2.T) - 1) / 2))
    return np.array([theta * 180 / np.pi, np.linalg.norm(RT1[:3, 3] - RT2[:3, 3]) * 100])
