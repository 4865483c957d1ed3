"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def synth_points(B, N, seed):
    """Same generator as tests/golden/make_golden.py::synth_points (SURVEY.md 8(d) inputs)."""
    g = torch.Generator().manual_seed(seed)
    pts = 0.1 * torch.randn(B, N, 3, generator=g)
    off = torch.rand(B, 1, 3, generator=g) * torch.tensor([0.4, 0.4, 1.0]) + torch.tensor([-0.2, -0.2, 0.5])
    obj = torch.randint(0, 6, (B, 1), generator=g).float()
    return (pts + off).contiguous(), obj


def knn_rows_equivalent(D, idx_a, idx_b):
    """Tie-aware comparison of two neighbour lists for the same distance matrix.

    D (n,n) fp32 distances (the pinned definition), idx_* (n,k).  torch.topk leaves the order of
    equal distances undefined, so the reference's list may differ from the (distance, index)
    policy exactly where distances tie.  Returns (exact_rows, multiset_rows): rows whose index
    lists are identical, and rows whose selected DISTANCE multisets are identical bit for bit.
    """
    idx_a = np.asarray(idx_a, dtype=np.int64)
    idx_b = np.asarray(idx_b, dtype=np.int64)
    exact = (idx_a == idx_b).all(axis=1)
    da = np.sort(np.take_along_axis(D, idx_a, axis=1), axis=1)
    db = np.sort(np.take_along_axis(D, idx_b, axis=1), axis=1)
    same = (da.view(np.int32) == db.view(np.int32)).all(axis=1)
    return exact, same


def rows_without_ties(D, k):
    """Rows whose k+2 smallest distances are pairwise distinct (no tie can affect a top-(k+1))."""
    s = np.sort(D, axis=1)[:, : k + 2]
    return (np.diff(s, axis=1) > 0).all(axis=1)


def synth_eval_results(seed, n_img=40):
    """A synthetic ``final_results`` list in the evaluater's format (evaluater/RT_TDA_Evaluater.py:99-105): per image a few
    ground-truth instances of the six NOCS classes, predictions = perturbed ground truth + false positives + misses."""
    rng = np.random.RandomState(seed)

    def rand_RT(scale_noise=0.0):
        q = rng.randn(4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        RT = np.eye(4)
        RT[:3, :3] = R * (1.0 + scale_noise * rng.rand())
        RT[:3, 3] = rng.uniform(-0.3, 0.3, 3) + np.array([0, 0, 1.0])
        return RT

    def perturb(RT, ang, shift):
        ax = rng.randn(3)
        ax /= np.linalg.norm(ax)
        a = np.deg2rad(ang) * rng.rand()
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        dR = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
        out = RT.copy()
        out[:3, :3] = dR @ RT[:3, :3]
        out[:3, 3] += rng.randn(3) * shift
        return out

    results = []
    for img in range(n_img):
        G = rng.randint(0, 6) if img % 9 else 0
        gt_cls = rng.randint(1, 7, G).astype(np.int32)
        gt_RTs = np.stack([rand_RT(0.3) for _ in range(G)]) if G else np.zeros((0, 4, 4))
        gt_scales = rng.uniform(0.05, 0.4, (G, 3))
        hv = rng.randint(0, 2, G)
        pr_cls, pr_RT, pr_sc, pr_score = [], [], [], []
        for j in range(G):
            if rng.rand() < 0.85:
                pr_cls.append(gt_cls[j] if rng.rand() < 0.9 else rng.randint(1, 7))
                pr_RT.append(perturb(gt_RTs[j], rng.choice([3, 8, 25, 90]), rng.choice([0.005, 0.02, 0.08])))
                pr_sc.append(gt_scales[j] * rng.uniform(0.8, 1.25, 3))
                pr_score.append(rng.uniform(0.3, 1.0))
        for _ in range(rng.randint(0, 3)):
            pr_cls.append(rng.randint(1, 7)), pr_RT.append(rand_RT()), pr_sc.append(rng.uniform(0.05, 0.4, 3))
            pr_score.append(rng.uniform(0.05, 0.6))
        P = len(pr_cls)
        results.append(dict(gt_class_ids=gt_cls, gt_RTs=gt_RTs, gt_scales=gt_scales, gt_handle_visibility=hv,
                            pred_bboxes=rng.randint(1, 400, (P, 4)).astype(np.float64), pred_class_ids=np.array(pr_cls, dtype=np.int32),
                            pred_scales=np.array(pr_sc).reshape(P, 3), pred_scores=np.array(pr_score),
                            pred_RTs=np.stack(pr_RT) if P else np.zeros((0, 4, 4))))
    return results


def synth_loss_batch(seed=41, B=10, N=96, D=50, C=70):
    """A training-step's worth of loss operands on the CPU: predictions near their targets (so the smooth-L1 knee, the
    confidence target exp(-13.7 d^2) and the sign flips all get exercised), every symmetry pattern of datasets' sym_info plus
    the degenerate ones, persistence images with some all-zero rows.  -> pred dict, gt dict, sym (B,4) int64, extras"""
    import torch
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(B, 3, 3)
    t = 0.1 * torch.randn(B, 3, generator=g) + torch.tensor([0.0, 0.0, 0.8])
    s = 0.2 + 0.3 * torch.rand(B, 3, generator=g)
    pats = [[1, 1, 0, 1], [1, 1, 1, 1], [0, 0, 0, 0], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 1, 0], [1, 0, 1, 0]]
    sym = torch.tensor([pats[i % len(pats)] for i in range(B)], dtype=torch.int64)
    nrm = lambda v: v / v.norm(dim=1, keepdim=True)
    cano = (torch.rand(B, N, 3, generator=g) - 0.5) * s.unsqueeze(1)
    PC = cano @ R.transpose(1, 2) + t.unsqueeze(1)
    h = lambda: torch.relu(torch.randn(B, D, generator=g)) * (torch.rand(B, 1, generator=g) > 0.25)
    gt = {"Rot1": R[:, :, 1].contiguous(), "Rot2": R[:, :, 0].contiguous(), "Recon": PC, "Tran": t, "Size": s, "h1": h(), "h2": h(),
          "pdh1_category": h(), "pdh2_category": h(), "R": R}
    pred = {"Rot1": nrm(gt["Rot1"] + 0.15 * torch.randn(B, 3, generator=g)), "Rot2": nrm(gt["Rot2"] + 0.15 * torch.randn(B, 3, generator=g)),
            "Rot1_f": torch.rand(B, generator=g), "Rot2_f": torch.rand(B, generator=g),
            "Recon": PC + 0.02 * torch.randn(B, N, 3, generator=g), "Tran": t + 0.3 * torch.randn(B, 3, generator=g),
            "Size": s + 0.3 * torch.randn(B, 3, generator=g), "TDA_h1": gt["h1"] + 0.3 * torch.randn(B, D, generator=g),
            "TDA_h2": gt["h2"] + 0.3 * torch.randn(B, D, generator=g)}
    extra = {"feat1": torch.randn(B, C, generator=g), "feat2": torch.randn(B, C, generator=g),
             "recon2": PC + 0.02 * torch.randn(B, N, 3, generator=g)}
    extra["feat2"][3] = 0.0                                # a row F.normalize clamps
    return pred, gt, sym, extra


def rand_rotations(B, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                        2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(B, 3, 3)


def synth_train_db(cat_points, cat_h1, cat_h2, sym_table, cat_ids, N, seed):
    """A batch dict in the train loader's format (datasets/load_data.py:313-349) built from per-category tables: cat_points
    (6,1024,3), cat_h1 / cat_h2 (6,2500), sym_table 6 x 4.  Clouds are posed, scaled, noisy samples of the category clouds;
    aug_pcl_in is the same object jittered and slightly moved; pdh1 / pdh2 the category priors with noise, clipped to [0, 1]."""
    import torch
    g = torch.Generator().manual_seed(seed)
    B = len(cat_ids)
    cid = torch.tensor(cat_ids)
    R = rand_rotations(B, seed + 1)
    t = torch.randn(B, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 0.9])
    s = torch.rand(B, 3, generator=g) * 0.1 + 0.2
    P = cat_points.shape[1]                       # N <= P: distinct points; beyond that the surplus is drawn with replacement
    pick = torch.stack([torch.cat([torch.randperm(P, generator=g), torch.randint(0, P, (max(N - P, 0),), generator=g)])[:N]
                        for _ in range(B)])                                                                  # (B, N)
    cano = torch.gather(cat_points[cid], 1, pick.unsqueeze(-1).expand(B, N, 3))
    pcl = torch.matmul(cano * s.unsqueeze(1), R.transpose(1, 2)) + t.unsqueeze(1) + 0.002 * torch.randn(B, N, 3, generator=g)
    dR = rand_rotations(B, seed + 2)
    dR = torch.eye(3) + 0.05 * (dR - dR.transpose(1, 2))                                                    # a small motion
    ctr = pcl.mean(1, keepdim=True)
    aug = torch.matmul(pcl - ctr, dR.transpose(1, 2)) + ctr + 0.01 * torch.randn(B, 1, 3, generator=g) + 0.003 * torch.randn(B, N, 3, generator=g)
    noisy = lambda h: (h[cid] + 0.05 * torch.randn(B, h.shape[1], generator=g)).clamp(0, 1)
    return {"pcl_in": pcl.contiguous(), "aug_pcl_in": aug.contiguous(), "cat_id": cid.float().view(B, 1), "rotation": R, "translation": t,
            "fsnet_scale": s, "sym_info": torch.tensor([sym_table[c] for c in cat_ids], dtype=torch.float32),
            "pdh1": noisy(cat_h1), "pdh2": noisy(cat_h2), "pdh1_category": cat_h1[cid].clone(), "pdh2_category": cat_h2[cid].clone(),
            "points_category": cat_points[cid].clone()}


def synth_depth_scene(seed, n_det=4, H=480, W=640, edge_cases=False):
    """A synthetic frame in the layout the evaluation loader reads (evaluation/load_data_eval.py:271-303): a uint16 depth
    image in millimetres (sloped background, nearer blobs, zero-depth holes), Mask-RCNN style ``pred_masks`` (H,W,n) bool,
    ``pred_bboxes`` (n,4) int32 as (y1,x1,y2,x2), ``pred_class_ids`` (n,) in 1..6.  ``edge_cases`` adds a box hanging over
    the image border, a tiny mask (fewer than 1024 ROI points: the tiling branch) and a box larger than get_bbox's 440 cap."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    depth = 1400.0 + 0.4 * xx + 0.25 * yy + rng.randn(H, W) * 3.0
    masks = np.zeros((H, W, n_det), dtype=bool)
    boxes = np.zeros((n_det, 4), dtype=np.int32)
    for j in range(n_det):
        ry, rx = rng.randint(30, 110), rng.randint(30, 110)
        cy, cx = rng.randint(60, H - 60), rng.randint(60, W - 60)
        if edge_cases and j == 0:
            cy, cx = 20, W - 25                                 # window pushed back inside the image by get_bbox
        if edge_cases and j == 1:
            ry, rx = 2, 3                                       # ~20 source pixels, ~800 ROI points -> tiled up to n_pts
        if edge_cases and j == 2:
            ry, rx, cy, cx = 235, 300, H // 2, W // 2           # wider than the 440 window cap
        e = ((yy - cy) / float(ry)) ** 2 + ((xx - cx) / float(rx)) ** 2
        m = e <= 1.0
        masks[:, :, j] = m
        depth = np.where(m, 700.0 + 60.0 * j + 90.0 * np.sqrt(np.clip(1.0 - e, 0, 1)) * -1.0 + rng.randn(H, W) * 1.5, depth)
        ys, xs = np.where(m)
        boxes[j] = [ys.min(), xs.min(), ys.max() + 1, xs.max() + 1]
    holes = rng.rand(H, W) < 0.03
    depth = np.where(holes, 0.0, depth)
    depth[:, : W // 40] = 0.0                                   # a dead band, as structured-light sensors have
    cls = rng.randint(1, 7, n_det).astype(np.int32)
    return dict(depth=np.clip(depth, 0, 65535).astype(np.uint16), pred_masks=masks, pred_bboxes=boxes, pred_class_ids=cls,
                pred_scores=rng.uniform(0.5, 1.0, n_det))
