"""Oracle (TEST INFRASTRUCTURE): the input side of the evaluation loader restated in numpy -- depth image + detection mask +
box -> the (n_pts, 3) camera-frame cloud the network is fed (evaluation/load_data_eval.py:294-357, 404-417, 451-462;
tools/dataset_utils.py:80-136; get_bbox as network/point_sample/pc_sample_sphere.py:456-484).

Pinning.  Everything except the two OpenCV calls is pinned by tests/golden/input_side.npz, which make_golden.py records by
running the reference's own ``PoseDataset.__getitem__`` on synthetic scenes.  ``cv2`` is not installed in the build image
(and cannot be), so ``cv2.getAffineTransform`` / ``cv2.warpAffine(..., INTER_NEAREST)`` are restated here from OpenCV's
published algorithm (imgproc/src/imgwarp.cpp: 6x6 solve in double, matrix inversion in double, 10-bit fixed-point source
coordinates with round_delta = 512, constant zero border): **parity of those two functions is unpinned**; the golden run
uses this restatement as its ``cv2`` stand-in.
"""
import numpy as np

AB_BITS = 10
AB_SCALE = 1 << AB_BITS


def get_bbox(bbox):
    """pc_sample_sphere.py:456-484 (the source twin of tools.eval_utils.get_bbox, which ships as bytecode only)."""
    y1, x1, y2, x2 = bbox
    img_width, img_length = 480, 640
    window_size = (max(y2 - y1, x2 - x1) // 40 + 1) * 40
    window_size = min(window_size, 440)
    center = [(y1 + y2) // 2, (x1 + x2) // 2]
    rmin = center[0] - int(window_size / 2)
    rmax = center[0] + int(window_size / 2)
    cmin = center[1] - int(window_size / 2)
    cmax = center[1] + int(window_size / 2)
    if rmin < 0:
        rmax += -rmin
        rmin = 0
    if cmin < 0:
        cmax += -cmin
        cmin = 0
    if rmax > img_width:
        rmin -= rmax - img_width
        rmax = img_width
    if cmax > img_length:
        cmin -= cmax - img_length
        cmax = img_length
    return rmin, rmax, cmin, cmax


def get_affine_transform_cv(src, dst):
    """cv2.getAffineTransform: solve the 6x6 system for the 2x3 matrix taking three src points to three dst points (double)."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:2], A[2 * i, 2] = src[i], 1.0
        A[2 * i + 1, 3:5], A[2 * i + 1, 5] = src[i], 1.0
        b[2 * i], b[2 * i + 1] = dst[i]
    return np.linalg.solve(A, b).reshape(2, 3)


def nearest_source_map(M, dsize):
    """cv2.warpAffine's source pixel per destination pixel for INTER_NEAREST (no WARP_INVERSE_MAP): invert M in double, then
    X = (round((M1*y + M2)*1024) + 512 + round(M0*x*1024)) >> 10, likewise Y.  -> (sx, sy) int arrays of shape (h, w)."""
    w, h = int(dsize[0]), int(dsize[1])
    M = np.asarray(M, dtype=np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * -D, M[1, 0] * -D, A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    rnd = lambda v: np.rint(v).astype(np.int64)          # cvRound: to nearest, ties to even
    adelta, bdelta = rnd(M[0, 0] * x * AB_SCALE), rnd(M[1, 0] * x * AB_SCALE)
    X0 = rnd((M[0, 1] * y + M[0, 2]) * AB_SCALE) + AB_SCALE // 2
    Y0 = rnd((M[1, 1] * y + M[1, 2]) * AB_SCALE) + AB_SCALE // 2
    sx = (X0[:, None] + adelta[None, :]) >> AB_BITS
    sy = (Y0[:, None] + bdelta[None, :]) >> AB_BITS
    return np.clip(sx, -32768, 32767), np.clip(sy, -32768, 32767)       # saturate_cast<short>


def warp_affine_nearest(img, M, dsize):
    """cv2.warpAffine(img, M, dsize, flags=cv2.INTER_NEAREST), BORDER_CONSTANT 0; img (H,W) or (H,W,C)."""
    sx, sy = nearest_source_map(M, dsize)
    H, W = img.shape[:2]
    inb = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    out = np.zeros((sx.shape[0], sx.shape[1]) + img.shape[2:], dtype=img.dtype)
    out[inb] = img[sy[inb], sx[inb]]
    return out


def roi_affine(bbox_center, scale, out_size):
    """tools/dataset_utils.py:95-136 with rot = 0, shift = 0 (get_affine_transform) -> 2x3 matrix."""
    center = np.asarray(bbox_center)
    src_w, dst_w, dst_h = scale, out_size, out_size
    src_dir = [0 * 1.0 - (src_w * -0.5) * 0.0, 0 * 0.0 + (src_w * -0.5) * 1.0]          # get_dir(.., rot_rad=0)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = center + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    third = lambda a, b: b + np.array([-(a - b)[1], (a - b)[0]], dtype=np.float32)
    src[2, :] = third(src[0], src[1])
    dst[2, :] = third(dst[0], dst[1])
    return get_affine_transform_cv(src, dst)


def roi_source_map(bbox, im_H, im_W, img_size=256):
    """load_data_eval.py:305-316: detection box -> square window -> source pixel of each ROI pixel."""
    rmin, rmax, cmin, cmax = get_bbox(bbox)
    x1, y1, x2, y2 = cmin, rmin, cmax, rmax
    cx, cy = 0.5 * (x1 + x2), 0.5 * (y1 + y2)
    scale = min(max(y2 - y1, x2 - x1), max(im_H, im_W)) * 1.0
    return nearest_source_map(roi_affine(np.array([cx, cy]), scale, img_size), (img_size, img_size))


def depth_to_pcl(depth, K, xymap, mask):
    """load_data_eval.py:451-462, float32 throughout (K is a float32 array in the reference, :158-161)."""
    K = np.asarray(K, dtype=np.float32).reshape(-1)
    cx, cy, fx, fy = K[2], K[5], K[0], K[4]
    depth = depth.reshape(-1).astype(np.float32)
    valid = ((depth > 0) * mask.reshape(-1)) > 0
    depth = depth[valid]
    real_x = (xymap[0].reshape(-1)[valid] - cx) * depth / fx
    real_y = (xymap[1].reshape(-1)[valid] - cy) * depth / fy
    return np.stack((real_x, real_y, depth), axis=-1).astype(np.float32)


def roi_cloud(depth, mask, bbox, K, img_size=256):
    """One detection up to (not including) the resampling: load_data_eval.py:302-355.
    -> (cloud (m,3) float32 after the outlier cut, n_depth_valid, n_valid).  Returns cloud None where the reference's
    __getitem__ returns None (:332-337); raises IndexError where the reference does (fewer than 26 valid points, :350)."""
    im_H, im_W = depth.shape
    sx, sy = roi_source_map(bbox, im_H, im_W, img_size)
    inb = (sx >= 0) & (sx < im_W) & (sy >= 0) & (sy < im_H)
    sxc, syc = np.where(inb, sx, 0), np.where(inb, sy, 0)
    roi_depth = np.where(inb, depth[syc, sxc], 0)
    roi_mask = np.where(inb, mask[syc, sxc], 0).astype(np.float32)
    xymap = np.stack([np.where(inb, sx, 0), np.where(inb, sy, 0)]).astype(np.float32)      # warped coord_2d, zero border
    n_depth = int((roi_depth > 0).sum())
    n_valid = int(((roi_depth > 0) & (roi_mask != 0)).sum())
    if n_depth <= 1 or n_valid <= 1:
        return None, n_depth, n_valid
    pcl = depth_to_pcl(roi_depth, K, xymap, roi_mask) / 1000.0
    ranges = pcl.max(axis=0) - pcl.min(axis=0)
    diag = np.sqrt(np.sum(ranges ** 2))
    centre = pcl[np.array([25])]
    dist = np.linalg.norm(pcl - centre, axis=1)
    return pcl[dist > diag * 0.25], n_depth, n_valid


def sample_selection(total, n_pts, rng=np.random):
    """load_data_eval.py:404-417 as an index list: tile when short, a prefix of one permutation when long."""
    if total < n_pts:
        return np.arange(n_pts) % total          # ZeroDivisionError for an empty cloud, as in the reference
    if total > n_pts:
        return rng.permutation(total)[:n_pts]
    return np.arange(n_pts)


def image_clouds(depth, masks, bboxes, K, img_size=256, n_pts=1024, rng=np.random):
    """All detections of one image, in order, drawing from ``rng`` as __getitem__ does -> (n, n_pts, 3) or None."""
    out = []
    for j in range(len(bboxes)):
        pcl, _, _ = roi_cloud(depth, masks[:, :, j], bboxes[j], K, img_size)
        if pcl is None:
            return None
        out.append(pcl[sample_selection(len(pcl), n_pts, rng)])
    return np.array(out, dtype=np.float32).reshape(len(out), n_pts, 3)


# ---------------------------------------------------------------------------------------------------------------------------
# Training loader, the same stages with the ground-truth instance mask (datasets/load_data.py:232-290, 335-336, 366-380, 395-407).
# Augmentation (aug_bbox_DZI's random window, defor_2D, PC_BasicAugment, the custom operators) is the caller's: the window is an
# argument here, the point-cloud augmentations are identity.
def dzi_window_off(bbox, im_H, im_W):
    """load_data.py:235-238 with FLAGS.DZI_TYPE set to none of the augmenting kinds (tools/dataset_utils.py:57-61):
    get_bbox's window -> (bbox_center (cx, cy), scale)."""
    rmin, rmax, cmin, cmax = get_bbox(bbox)
    x1, y1, x2, y2 = cmin, rmin, cmax, rmax
    return np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)]), min(max(y2 - y1, x2 - x1), max(im_H, im_W)) * 1.0


def train_roi_cloud(depth, mask, inst_id, bbox_center, scale, K, img_size=256):
    """One training item up to the subsampling (load_data.py:239-290): ROI resampling of pixel grid / instance mask / depth with the
    given window, validity tests (:260-265), _depth_to_pcl (:395-407) / 1000, the cut of the points within 0.15 x the extent's
    diagonal of point number 25 (:276-286).  -> (cloud or None where __getitem__ retries, n_depth_valid, n_valid)."""
    im_H, im_W = depth.shape
    sx, sy = nearest_source_map(roi_affine(np.asarray(bbox_center), scale, img_size), (img_size, img_size))
    inb = (sx >= 0) & (sx < im_W) & (sy >= 0) & (sy < im_H)
    sxc, syc = np.where(inb, sx, 0), np.where(inb, sy, 0)
    roi_depth = np.where(inb, depth[syc, sxc], 0)
    mask_target = (mask == inst_id).astype(np.float32)                                  # :245-247
    roi_mask = np.where(inb, mask_target[syc, sxc], 0).astype(np.float32)
    xymap = np.stack([np.where(inb, sx, 0), np.where(inb, sy, 0)]).astype(np.float32)
    n_depth = int((roi_depth > 0).sum())
    n_valid = int(((roi_depth > 0) & (roi_mask != 0)).sum())
    if n_depth <= 1 or n_valid <= 1:
        return None, n_depth, n_valid
    pcl = depth_to_pcl(roi_depth, K, xymap, roi_mask) / 1000.0
    ranges = pcl.max(axis=0) - pcl.min(axis=0)
    diag = np.sqrt(np.sum(ranges ** 2))
    dist = np.linalg.norm(pcl - pcl[np.array([25])], axis=1)
    pcl = pcl[dist > (diag * 0.15)]                                                     # float32 product (NumPy 2 promotion)
    return (pcl if len(pcl) >= 50 else None), n_depth, n_valid                          # :288-289


def train_item_clouds(depth, mask, inst_id, bbox_center, scale, K, img_size=256, rng=np.random):
    """-> (PC (2048,3), pcl_in (1024,3)) as load_data.py:335-336 draws them (two _sample_points calls, :366-380), or None."""
    pcl, _, _ = train_roi_cloud(depth, mask, inst_id, bbox_center, scale, K, img_size)
    if pcl is None:
        return None
    PC = pcl[sample_selection(len(pcl), 2048, rng)]
    return PC, PC[sample_selection(2048, 1024, rng)]
