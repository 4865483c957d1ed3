"""Input side of the evaluation loader on the device.

Stands where the per-detection loop of ``PoseDataset.__getitem__`` stands in the reference
(``evaluation/load_data_eval.py:294-357``): for every Mask-RCNN detection of a frame it crops a square window round the
box, resamples depth / mask / pixel grid to ``FLAGS.img_size`` squared with nearest-neighbour ``cv2.warpAffine``,
back-projects the valid pixels (``_depth_to_pcl`` :451-462), cuts the points near point number 25 (:341-355) and
resamples to ``FLAGS.random_points`` (``_sample_points`` :404-417) -- about ten numpy / OpenCV passes over 65536 pixels per
detection on a DataLoader worker.  Here the frames' depth images and masks are uploaded once and ONE launch
(``tgp_roi_cloud``, a workgroup per detection) leaves every detection's cloud in HBM, in the reference's point order, as
4-byte records (ROI pixel, depth); the resampling is a gather that materialises only the 1024 selected points.  The clouds never visit the host: they are the ``pcl_in`` that ``PoseNet9D.forward`` consumes
(``pose.batched_inference``).

``sampler='numpy'`` reproduces the reference bit for bit, including its use of the global ``np.random`` stream: the
per-detection point counts (12 bytes each) are read back once per call, the permutations are drawn on the host in the
reference's order (frame by frame, detection by detection; a frame the reference abandons with ``return None`` has consumed
the draws of its earlier detections, as there), and their prefixes are uploaded for the gather.
``sampler='device'`` draws a keyed pseudo-random permutation on the device instead (``tgp_cloud_sample``): nothing is
read back, frames cannot be dropped on the host, so invalid detections come back as NaN rows with ``valid`` False.

What stays on the host, as integer arithmetic on four numbers per detection: ``get_bbox`` (the window rule of
``tools.eval_utils``, source twin ``network/point_sample/pc_sample_sphere.py:456-484``).  File reading / unpickling
(:239-284) and the category bookkeeping (:359-400) are not part of this module.
"""
import numpy as np
import torch

from .. import ops

REAL_INTRINSICS = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]], dtype=np.float32)   # :160
CAMERA_INTRINSICS = np.array([[577.5, 0, 319.5], [0, 577.5, 239.5], [0, 0, 1]], dtype=np.float32)              # :158


def get_bbox(bbox):
    """Square crop window of a detection box (y1, x1, y2, x2) -> rmin, rmax, cmin, cmax (pc_sample_sphere.py:456-484)."""
    y1, x1, y2, x2 = (int(v) for v in bbox)
    img_width, img_length = 480, 640
    window_size = min((max(y2 - y1, x2 - x1) // 40 + 1) * 40, 440)
    half = int(window_size / 2)
    cy, cx = (y1 + y2) // 2, (x1 + x2) // 2
    rmin, rmax, cmin, cmax = cy - half, cy + half, cx - half, cx + half
    if rmin < 0:
        rmin, rmax = 0, rmax - rmin
    if cmin < 0:
        cmin, cmax = 0, cmax - cmin
    if rmax > img_width:
        rmin, rmax = rmin - (rmax - img_width), img_width
    if cmax > img_length:
        cmin, cmax = cmin - (cmax - img_length), img_length
    return rmin, rmax, cmin, cmax


def _windows(frames):
    """Host packing: per detection {cmin+cmax, rmin+rmax, s} (:305-316), its frame, its mask channel's offset and stride."""
    win, det_img, off, stride = [], [], [], []
    pos = 0
    for i, fr in enumerate(frames):
        H, W, n = fr["pred_masks"].shape
        if n >= 128:
            raise ValueError("frame %d: %d mask channels; tgp_roi_cloud addresses masks with 31-bit offsets (n < 128)" % (i, n))
        if len(fr["pred_bboxes"]) != n:
            raise ValueError("frame %d: %d boxes for %d mask channels" % (i, len(fr["pred_bboxes"]), n))
        for j in range(n):
            rmin, rmax, cmin, cmax = get_bbox(fr["pred_bboxes"][j])
            s = min(max(rmax - rmin, cmax - cmin), max(H, W))
            win.append((cmin + cmax, rmin + rmax, s))
            det_img.append(i), off.append(pos + j), stride.append(n)
        pos += H * W * n
    return win, det_img, off, stride


class RoiClouds:
    """Device-side result of ``build``: ``records`` (ops.RoiRecords: 4-byte records + counts (D,3) + descriptors), detections per
    frame.  ``points(n)`` materialises the first n points of every cloud (rows beyond a cloud's count are meaningless)."""

    def __init__(self, records, per_frame):
        self.records, self.per_frame = records, per_frame
        self.counts = None if records is None else records.counts

    def points(self, n):
        D = self.records.recs.shape[0]
        sel = torch.arange(n, dtype=torch.int32, device=self.records.recs.device).repeat(D, 1)
        return ops.cloud_select(self.records, sel)


def upload(frames, camK=REAL_INTRINSICS, device="cuda"):
    """Pack and upload what tgp_roi_cloud reads: (depth, masks, mask_off, mask_stride, det_img, window, camk) on the device."""
    dev = torch.device(device)
    H, W = frames[0]["depth"].shape
    for fr in frames:
        if fr["depth"].shape != (H, W) or fr["pred_masks"].shape[:2] != (H, W) or fr["depth"].dtype != np.uint16:
            raise ValueError("every frame needs a uint16 depth image and masks of one common (H,W)")
    win, det_img, off, stride = _windows(frames)
    camK = np.asarray(camK, dtype=np.float32)
    camK = np.broadcast_to(camK, (len(frames), 3, 3)) if camK.ndim == 2 else camK
    camk = np.stack([camK[:, 0, 0], camK[:, 1, 1], camK[:, 0, 2], camK[:, 1, 2]], axis=1).astype(np.float32)
    # The copies go out on their own stream: a host-to-device copy from pageable memory holds the host until it has run, and on
    # the compute stream it would queue behind the previous chunk's forward -- the host would pack chunk c+1 only after chunk c
    # had finished.  The compute stream waits for the copies' event; record_stream keeps the allocator from recycling the
    # buffers before the compute stream is done with them.
    cur = torch.cuda.current_stream(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    cs = _COPY_STREAMS[key]
    host = [np.stack([fr["depth"] for fr in frames]).view(np.int16),
            np.concatenate([np.ascontiguousarray(fr["pred_masks"]).view(np.uint8).reshape(-1) for fr in frames if fr["pred_masks"].size]),
            np.asarray(off, dtype=np.int64), np.asarray(stride, dtype=np.int32), np.asarray(det_img, dtype=np.int32),
            np.asarray(win, dtype=np.int32).reshape(-1, 3), camk]
    with torch.cuda.stream(cs):
        out = [torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True) for a in host]
    done = torch.cuda.Event()
    done.record(cs)
    cur.wait_event(done)
    for t in out:
        t.record_stream(cur)
    return tuple(out)


_COPY_STREAMS = {}


def build(frames, camK=REAL_INTRINSICS, img_size=256, device="cuda"):
    """Upload the frames and run the per-detection kernel.  frames: list of dicts with 'depth' (H,W) uint16,
    'pred_masks' (H,W,n) bool / uint8, 'pred_bboxes' (n,4) (the detection pickle's layout, :271-303); camK one (3,3)
    matrix or one per frame.  All frames must share (H,W).  Nothing synchronises."""
    dev = torch.device(device)
    per_frame = [fr["pred_masks"].shape[2] for fr in frames]
    if sum(per_frame) == 0:
        return RoiClouds(None, per_frame)
    return RoiClouds(ops.roi_cloud(*upload(frames, camK, dev), roi_size=img_size), per_frame)


def clouds_from_frames(frames, camK=REAL_INTRINSICS, img_size=256, n_pts=1024, sampler="numpy", rng=np.random, seed=0, device="cuda"):
    """-> list over frames of ``pcl_in`` (n_det, n_pts, 3) float32 GPU tensors; ``None`` for a frame the reference's
    ``__getitem__`` drops (:332-337).  Raises IndexError / ZeroDivisionError where the reference does (fewer than 26 valid
    points :350, an empty cloud after the cut :411).  With sampler='device' returns (list of tensors, list of bool masks)."""
    rc = build(frames, camK, img_size, device)
    D = sum(rc.per_frame)
    dev = torch.device(device)
    empty = torch.zeros(0, n_pts, 3, device=dev)
    if sampler == "device":
        if D == 0:
            return [empty for _ in rc.per_frame], [torch.zeros(0, dtype=torch.bool, device=dev) for _ in rc.per_frame]
        out = ops.cloud_sample(rc.records, n_pts, seed)
        ok = (rc.counts[:, 2] > 0) & (rc.counts[:, 0] > 1) & (rc.counts[:, 1] > 1)
        return list(out.split(rc.per_frame)), list(ok.split(rc.per_frame))
    if sampler != "numpy":
        raise ValueError("sampler must be 'numpy' or 'device'")
    if D == 0:
        return [empty for _ in rc.per_frame]
    counts = rc.counts.cpu().numpy()                         # the one read-back: 12 bytes per detection
    sel = np.zeros((D, n_pts), dtype=np.int32)
    keep_frame, d = [], 0
    for n in rc.per_frame:
        alive = True
        for j in range(n):
            n_depth, n_valid, total = (int(v) for v in counts[d + j])
            if n_depth <= 1 or n_valid <= 1:                 # :332-337 -> return None (earlier detections already drew)
                alive = False
                break
            if total < 0:
                raise IndexError("index 25 is out of bounds for axis 0 with size %d" % n_valid)     # :350
            if total < n_pts:
                sel[d + j] = np.arange(n_pts) % total        # ZeroDivisionError for total == 0, as n_pts // 0 at :411
            elif total > n_pts:
                sel[d + j] = rng.permutation(total)[:n_pts]
            else:
                sel[d + j] = np.arange(n_pts)
        keep_frame.append(alive)
        d += n
    out = ops.cloud_select(rc.records, torch.from_numpy(sel).to(dev))
    return [c if alive else None for c, alive in zip(out.split(rc.per_frame), keep_frame)]
