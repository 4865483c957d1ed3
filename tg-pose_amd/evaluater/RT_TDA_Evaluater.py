"""The evaluation driver (``evaluater/RT_TDA_Evaluater.py``: ``myEvaluater.run`` :55-109, ``calc_pose_metric`` :111-175) over the
device pipeline: frames -> clouds (``evaluation.load_data_eval``) -> ``PoseNet9D.forward`` -> pose assembly
(``pose``, f-1) -> NOCS mAP (``evaluation.metrics``, f-2).

The reference walks the dataset one image at a time: ``__getitem__`` builds the clouds on the CPU, ``run`` uploads them, runs a
forward of a handful of objects, assembles the poses and copies them back -- two synchronisations and a dozen tiny launches per
image.  ``run`` here takes the same per-image records minus the CPU cloud building (depth image + detection pickle + ground
truth) in chunks of ``frames_per_batch`` images: one ``tgp_roi_cloud`` launch, forwards of up to ``max_batch`` objects, one
device-to-host copy of the (n,4,4) / (n,3) results per chunk.  It returns the list the reference pickles
(``evaluation/evaluate.py:53-67``): the detection dict without ``pred_masks`` plus ``pred_RTs`` / ``pred_scales``, merged
with the ground-truth record.

The per-category tables are the reference's data (``evaluation/load_data_eval.py:477-545`` mean shapes in millimetres,
``:547-566`` symmetry flags); file reading stays with the caller.
"""
import numpy as np
import torch

from ..evaluation import load_data_eval as lde
from ..evaluation.metrics import compute_degree_cm_mAP
from ..pose import batched_inference

SYNSET_NAMES = ['BG', 'bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']                                   # :115
MEAN_SHAPE_MM = {1: (87, 220, 89), 2: (165, 80, 165), 3: (88, 128, 156), 4: (68, 146, 72), 5: (346, 200, 335), 6: (146, 83, 114)}
SYM_INFO = {1: (1, 1, 0, 1), 2: (1, 1, 0, 1), 3: (0, 0, 0, 0), 4: (1, 1, 1, 1), 5: (0, 1, 0, 0), 6: (0, 1, 0, 0)}


class myEvaluater:
    def __init__(self, net, frames_per_batch=32, max_batch=256, sampler="numpy", seed=0):
        self.net1 = net.eval()
        self.device = next(net.parameters()).device
        self.frames_per_batch, self.max_batch, self.sampler, self.seed = frames_per_batch, max_batch, sampler, seed

    def _chunk(self, records, camK):
        frames = [r["frame"] for r in records]
        if self.sampler == "numpy":
            clouds = lde.clouds_from_frames(frames, camK, device=self.device)
            alive = [c is not None for c in clouds]
        else:
            clouds, ok = lde.clouds_from_frames(frames, camK, sampler="device", seed=self.seed, device=self.device)
            alive = [bool(o.all()) for o in ok]             # one small read-back per chunk: a frame with an invalid detection is dropped
        kept = [i for i, a in enumerate(alive) if a]
        ids = [np.asarray(frames[i]["pred_class_ids"]).astype(np.int64) for i in kept]
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)
        cat = [t(c - 1).reshape(-1, 1) for c in ids]                                                       # cat_id_0base (:368)
        mean = [t([MEAN_SHAPE_MM[int(c)] for c in cs]).reshape(-1, 3) / 1000.0 for cs in ids]              # :362
        sym = [t([SYM_INFO[int(c)] for c in cs]).reshape(-1, 4) for cs in ids]
        with torch.no_grad():
            poses = batched_inference(self.net1, [clouds[i] for i in kept], cat, mean, sym, max_batch=self.max_batch)
        out = []
        for i, p in zip(kept, poses):
            det = {k: v for k, v in frames[i].items() if k not in ("pred_masks", "depth")}
            det.update(p)
            det.update(records[i].get("gts", {}))
            out.append(det)
        return out

    def run(self, dataset, camK=lde.REAL_INTRINSICS):
        """dataset: iterable of {'frame': {'depth','pred_masks','pred_bboxes','pred_class_ids','pred_scores'}, 'gts': {...}}
        (None entries are skipped, as the reference skips them :66-67) -> pred_results list."""
        results, chunk = [], []
        for rec in dataset:
            if rec is None:
                continue
            chunk.append(rec)
            if len(chunk) == self.frames_per_batch:
                results += self._chunk(chunk, camK)
                chunk = []
        if chunk:
            results += self._chunk(chunk, camK)
        return results


def calc_pose_metric(pred_results, output_path, per_obj=""):
    """:111-175 -- the mAP tables and the lines the reference logs (degree 0..60, shift 0..10 cm by 0.5, IoU 0..1 by 0.01)."""
    degree, shift, iou = list(range(0, 61, 1)), [i / 2 for i in range(21)], [i / 100 for i in range(101)]
    idx = SYNSET_NAMES.index(per_obj) if per_obj in SYNSET_NAMES else -1
    iou_aps, pose_aps = compute_degree_cm_mAP(pred_results, SYNSET_NAMES, output_path, degree, shift, iou, iou_pose_thres=0.1,
                                              use_matches_for_pose=True)[:2]
    i25, i50, i75 = iou.index(0.25), iou.index(0.5), iou.index(0.75)
    d5, d10, s2, s5, s10 = degree.index(5), degree.index(10), shift.index(2), shift.index(5), shift.index(10)
    messages = ['mAP:' if idx != -1 else 'average mAP:',
                '3D IoU at 25: {:.1f}'.format(iou_aps[idx, i25] * 100), '3D IoU at 50: {:.1f}'.format(iou_aps[idx, i50] * 100),
                '3D IoU at 75: {:.1f}'.format(iou_aps[idx, i75] * 100),
                '5 degree, 2cm: {:.1f}'.format(pose_aps[idx, d5, s2] * 100), '5 degree, 5cm: {:.1f}'.format(pose_aps[idx, d5, s5] * 100),
                '10 degree, 2cm: {:.1f}'.format(pose_aps[idx, d10, s2] * 100), '10 degree, 5cm: {:.1f}'.format(pose_aps[idx, d10, s5] * 100),
                '10 degree, 10cm: {:.1f}'.format(pose_aps[idx, d10, s10] * 100),
                '5 degree: {:.1f}'.format(pose_aps[idx, d5, -1] * 100), '10 degree: {:.1f}'.format(pose_aps[idx, d10, -1] * 100),
                '2cm: {:.1f}'.format(pose_aps[idx, -1, s2] * 100), '5cm: {:.1f}'.format(pose_aps[idx, -1, s5] * 100),
                '10cm: {:.1f}'.format(pose_aps[idx, -1, s10] * 100)]
    return iou_aps, pose_aps, messages
