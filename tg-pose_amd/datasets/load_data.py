"""Device half of the TRAINING loader (the part round 2 left open).

Stands where ``PoseDataset.__getitem__`` of ``datasets/load_data.py`` does its per-item image work (:232-290, 335-336): the ROI
resampling of pixel grid / ground-truth instance mask / depth with nearest-neighbour ``cv2.warpAffine``
(``crop_resize_by_warp_affine``, tools/dataset_utils.py:80-136), the validity tests (:260-265), ``_depth_to_pcl`` (:395-407) / 1000,
the cut of the points within 0.15 x the extent's diagonal of point number 25 (:276-286), the ``len(pcl_in) < 50`` test (:288) and the
double subsample ``_sample_points(PC, 2048)`` then ``_sample_points(PC, 1024)`` (:335-336, 366-380).  A batch of items is ONE launch
of the evaluation loader's kernel (``tgp_roi_cloud_ex``: a workgroup per item, clouds kept as 4-byte records in the reference's
point order) plus one gather per requested cloud size (``tgp_cloud_select_ex``).

What differs from the evaluation loader, and how the kernel takes it:
  * the mask is the frame's instance-id image and the item is ``mask == inst_id`` (:245-247) -> ``mask_val``;
  * the window comes from ``aug_bbox_DZI`` (tools/dataset_utils.py:24-61), i.e. from the caller's augmentation draw: a real-valued
    centre and scale for which OpenCV's 10-bit fixed-point walk has no integer closed form -> ``source_tables`` evaluates that
    walk once per item on the host, in double as OpenCV does, and the kernel looks source pixels up;
  * the cut keeps the points farther than 0.15 (not 0.25) of the diagonal -> ``cut_frac``.

Out of scope, as in DESIGN.md section 8: reading files, the augmentations themselves (``aug_bbox_DZI``'s draw, ``defor_2D`` on the
mask, ``PC_BasicAugment`` and the custom operators on the cloud -- all identity / caller-supplied here), ``compute_pd``
(gudhi + persim).  With augmentation off the result is the reference's ``pcl_in`` bit for bit given the same ``np.random`` state
(tests/test_gpu_parity.py::test_train_loader_vs_reference_getitem)."""
import numpy as np
import torch

from .. import ops
from ..evaluation.load_data_eval import CAMERA_INTRINSICS, REAL_INTRINSICS, get_bbox  # noqa: F401  (re-exported)

AB_BITS = 10          # OpenCV's warpAffine works in 10-bit fixed point (AB_SCALE = 1024)


def window_without_dzi(bbox, im_H, im_W):
    """load_data.py:233-238 when FLAGS.DZI_TYPE names none of the augmenting kinds (tools/dataset_utils.py:57-61): get_bbox's window
    -> (bbox_center (cx, cy) float64, scale)."""
    rmin, rmax, cmin, cmax = get_bbox(bbox)
    return np.array([0.5 * (cmin + cmax), 0.5 * (rmin + rmax)]), min(max(rmax - rmin, cmax - cmin), max(im_H, im_W)) * 1.0


def source_tables(bbox_center, scale, img_size=256):
    """Source pixel of every ROI column and row: cv2.warpAffine(img, get_affine_transform(center, scale, 0, img_size), INTER_NEAREST).

    tools/dataset_utils.py:95-136 builds three float32 point pairs for rot = 0 -- (centre, centre - (0, scale / 2), and their
    perpendicular third) onto (size / 2, size / 2), ... -- and cv2.getAffineTransform solves for the 2 x 3 matrix in double; with
    rot = 0 that matrix is diag(size / scale) plus a shift, written down here from the same float32 points.  cv2.warpAffine inverts
    it in double and walks X = (round((M01 y + M02) 1024) + 512 + round(M00 x 1024)) >> 10 (likewise Y) with round-half-even.
    -> (2, img_size) int32: [0] source column per ROI column, [1] source row per ROI row."""
    c = np.asarray(bbox_center, dtype=np.float64)
    # the three point pairs, in the arithmetic types the reference's set-up uses (float32 arrays filled from float64 sums, the
    # perpendicular third point in float32): the matrix must agree with OpenCV's to the last bit of those points
    src, dst = np.zeros((3, 2), dtype=np.float32), np.zeros((3, 2), dtype=np.float32)
    src[0] = c
    src[1] = c + np.array([0.0, scale * -0.5])
    dst[0] = [img_size * 0.5, img_size * 0.5]
    dst[1] = np.array([img_size * 0.5, img_size * 0.5], np.float32) + np.array([0, img_size * -0.5], np.float32)
    perp = lambda a, b: b + np.array([-(a - b)[1], (a - b)[0]], dtype=np.float32)
    src[2], dst[2] = perp(src[0], src[1]), perp(dst[0], dst[1])
    pairs = [(src[i].astype(np.float64), dst[i].astype(np.float64)) for i in range(3)]
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i, (s, d) in enumerate(pairs):
        A[2 * i, 0:2], A[2 * i, 2] = s, 1.0
        A[2 * i + 1, 3:5], A[2 * i + 1, 5] = s, 1.0
        b[2 * i], b[2 * i + 1] = d
    M = np.linalg.solve(A, b).reshape(2, 3)
    # cv2.warpAffine without WARP_INVERSE_MAP: invert in double
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    i00, i01, i10, i11 = A11, M[0, 1] * -D, M[1, 0] * -D, A22
    b1 = -i00 * M[0, 2] - i01 * M[1, 2]
    b2 = -i10 * M[0, 2] - i11 * M[1, 2]
    scale_ab = float(1 << AB_BITS)
    x = np.arange(img_size, dtype=np.float64)
    rnd = lambda v: np.rint(v).astype(np.int64)
    half = 1 << (AB_BITS - 1)
    adelta, bdelta = rnd(i00 * x * scale_ab), rnd(i10 * x * scale_ab)               # per ROI column
    X0, Y0 = rnd((i01 * x + b1) * scale_ab) + half, rnd((i11 * x + b2) * scale_ab) + half      # per ROI row
    # rot = 0: the cross terms i01, i10 are the solver's rounding noise (~1e-17), so X0 does not depend on the row nor bdelta on the
    # column and the map is separable; a window for which that noise moves a rounding tie is refused rather than approximated
    sx2, sy2 = (X0[:, None] + adelta[None, :]) >> AB_BITS, (Y0[:, None] + bdelta[None, :]) >> AB_BITS      # OpenCV's map, all pixels
    sx, sy = sx2[0], sy2[:, 0]
    if not (np.array_equal(sx2, np.broadcast_to(sx[None, :], sx2.shape)) and np.array_equal(sy2, np.broadcast_to(sy[:, None], sy2.shape))):
        raise ValueError("source_tables: the walk is not separable for this window (rot = 0 expected)")
    return np.clip(np.stack([sx, sy]), -32768, 32767).astype(np.int32)       # (OpenCV stores the map as shorts)


def _selection(total, n_pts, rng):
    """_sample_points (:366-380) as indices: tile when short, the prefix of one permutation when long"""
    if total < n_pts:
        return np.arange(n_pts) % total
    if total > n_pts:
        return rng.permutation(total)[:n_pts]
    return np.arange(n_pts)


def train_clouds(items, img_size=256, rng=np.random, device="cuda", min_points=50):
    """items: list of dicts -- 'depth' (H,W) uint16 (load_depth's output), 'mask' (H,W) uint8 instance-id image (the reference reads
    ``cv2.imread(mask_path)[:, :, 2]``), 'inst_id' int, 'camK' (3,3) float32, and the window: 'bbox_center' (cx, cy) + 'scale' (the
    caller's aug_bbox_DZI draw) or 'bbox' (y1, x1, y2, x2) for the un-augmented window.  All frames share (H, W).
    -> list over items of (PC (2048,3), pcl_in (1024,3)) float32 GPU tensors, or None for an item the reference's __getitem__
    abandons and retries (:262-265 too few valid pixels, :288 fewer than 50 points after the cut).  The two permutations per item
    are drawn from ``rng`` in the reference's order (item by item, 2048 first).  One 12-byte read-back per item (the counts)."""
    dev = torch.device(device)
    if not items:
        return []
    H, W = items[0]["depth"].shape
    tabs, camk, mval = [], [], []
    for it in items:
        if it["depth"].shape != (H, W) or it["mask"].shape != (H, W) or it["depth"].dtype != np.uint16 or it["mask"].dtype != np.uint8:
            raise ValueError("every item needs a uint16 depth image and a uint8 instance mask of one common (H,W)")
        if not 0 < int(it["inst_id"]) < 256:
            raise ValueError("inst_id must be a non-zero byte value")
        center, scale = (it["bbox_center"], it["scale"]) if "bbox_center" in it else window_without_dzi(it["bbox"], H, W)
        tabs.append(source_tables(center, scale, img_size))
        K = np.asarray(it["camK"], dtype=np.float32)
        camk.append([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        mval.append(int(it["inst_id"]))
    D = len(items)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    depth = up(np.stack([it["depth"] for it in items]).view(np.int16))
    masks = up(np.stack([it["mask"] for it in items]).reshape(-1))
    rr = ops.roi_cloud(depth, masks, up(np.arange(D, dtype=np.int64) * (H * W)), up(np.ones(D, dtype=np.int32)),
                       up(np.arange(D, dtype=np.int32)), None, up(np.asarray(camk, dtype=np.float32)), roi_size=img_size,
                       tables=up(np.stack(tabs)), mask_val=up(np.asarray(mval, dtype=np.int32)), cut_frac=0.15)
    counts = rr.counts.cpu().numpy()
    sel2k, sel1k = np.zeros((D, 2048), dtype=np.int32), np.zeros((D, 1024), dtype=np.int32)
    alive = []
    for d in range(D):
        n_depth, n_valid, total = (int(v) for v in counts[d])
        if n_depth <= 1 or n_valid <= 1:                       # :262-265
            alive.append(False)
            continue
        if total < 0:
            raise IndexError("index 25 is out of bounds for axis 0 with size %d" % n_valid)     # :281
        if total < min_points:                                 # :288
            alive.append(False)
            continue
        sel2k[d] = _selection(total, 2048, rng)                # PC = _sample_points(PC, 2048)
        sel1k[d] = sel2k[d][_selection(2048, 1024, rng)]       # pcl_in = _sample_points(PC, 1024): a selection of the selection
        alive.append(True)
    pc2k = ops.cloud_select(rr, up(sel2k))
    pc1k = ops.cloud_select(rr, up(sel1k))
    return [(pc2k[d], pc1k[d]) if alive[d] else None for d in range(D)]
