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
import os

import numpy as np
import torch

from ..evaluation import load_data_eval as lde
from ..evaluation.metrics import compute_degree_cm_mAP
from ..pose import infer_device

SYNSET_NAMES = ['BG', 'bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']                                   # :115
MEAN_SHAPE_MM = {1: (87, 220, 89), 2: (165, 80, 165), 3: (88, 128, 156), 4: (68, 146, 72), 5: (346, 200, 335), 6: (146, 83, 114)}
SYM_INFO = {1: (1, 1, 0, 1), 2: (1, 1, 0, 1), 3: (0, 0, 0, 0), 4: (1, 1, 1, 1), 5: (0, 1, 0, 0), 6: (0, 1, 0, 0)}


class myEvaluater:
    """``sampler='numpy'``: the reference's draws (one small read-back per chunk for the point counts).  ``sampler='device'``:
    nothing is read back until a chunk's poses are; with ``overlap`` (default) chunk c's results are fetched after chunk c+1
    has been enqueued, so packing / uploading the next frames runs beside the GPU's work on the current ones."""

    def __init__(self, net, frames_per_batch=32, max_batch=256, sampler="numpy", seed=0, overlap=True, graph=False):
        self.net1 = net.eval()
        # The driver reads the six pose outputs only (:143-150): PH predictor and decoder are dead code here, as in the reference's
        # eval dict, so its forwards run lean BY DEFAULT (TGP_EVAL_FULL_FORWARD=1 or eval_outputs_only=False: the full forward).
        # The setting is this evaluater's and travels with each call; the caller's module is not touched.
        self.eval_outputs_only = not os.environ.get("TGP_EVAL_FULL_FORWARD")
        self.device = next(net.parameters()).device
        self.frames_per_batch, self.max_batch, self.sampler, self.seed, self.overlap = frames_per_batch, max_batch, sampler, seed, overlap
        self.graph = graph                                   # replay the forward as a captured hipGraph (PoseNet9D.graph_replay)
        self._fetch = torch.cuda.Stream(device=self.device)  # results come back on their own stream (see _finish)

    def _launch(self, records, camK):
        """Enqueue everything for a chunk of frames; returns what _finish needs.  No synchronisation on the device-sampler path."""
        frames = [r["frame"] for r in records]
        per = [fr["pred_masks"].shape[2] for fr in frames]
        if self.sampler == "numpy":
            clouds = lde.clouds_from_frames(frames, camK, device=self.device)
            alive, ok = [c is not None for c in clouds], None
        else:
            clouds, ok = lde.clouds_from_frames(frames, camK, sampler="device", seed=self.seed, device=self.device)
            alive = [True] * len(frames)                     # decided in _finish from `ok`
            # an invalid detection's rows are NaN: zero them for the forward (objects are independent in eval mode) and drop
            # the frame afterwards, as the reference drops it (load_data_eval.py:332-337)
        kept = [i for i, a in enumerate(alive) if a and per[i] > 0]
        if not kept:
            return records, alive, ok, kept, None, None, None
        ids = [np.asarray(frames[i]["pred_class_ids"]).astype(np.int64) for i in kept]
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device, non_blocking=True)
        flat = np.concatenate(ids)
        cat = t(flat - 1).reshape(-1, 1)                                                                   # cat_id_0base (:368)
        mean = t([MEAN_SHAPE_MM[int(c)] for c in flat]).reshape(-1, 3) / 1000.0                            # :362
        sym = t([SYM_INFO[int(c)] for c in flat]).reshape(-1, 4)
        pts = torch.cat([clouds[i] for i in kept])
        if ok is not None:
            pts = torch.nan_to_num(pts, nan=0.0)
        with torch.no_grad():
            rts, scales = infer_device(self.net1, pts, cat, mean, sym, self.max_batch, eval_outputs_only=self.eval_outputs_only)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        return records, alive, ok, kept, rts, scales, done

    def _finish(self, launched):
        records, alive, ok, kept, rts, scales, done = launched
        # The copies back are stream-ordered: on the compute stream they would queue behind the NEXT chunk's forward, which
        # has already been enqueued, and the overlap would be lost.  They run on a fetch stream that waits only for this
        # chunk's event.
        with torch.cuda.stream(self._fetch):
            if done is not None:
                self._fetch.wait_event(done)
            if ok is not None and len(ok):
                flags = torch.stack([o.all() if o.numel() else torch.ones((), dtype=torch.bool, device=self.device) for o in ok])
                flags.record_stream(self._fetch)
                okc = flags.cpu().tolist()                   # first read-back of the chunk (device sampler)
                alive = [a and bool(b) for a, b in zip(alive, okc)]
            if rts is not None:
                rts.record_stream(self._fetch), scales.record_stream(self._fetch)
                rts, scales = rts.cpu().numpy(), scales.cpu().numpy()
        out = []
        pos = 0
        empty = dict(pred_RTs=np.zeros((0, 4, 4)), pred_scales=np.zeros((0, 4, 4)))                       # RT_TDA_Evaluater.py:70-71
        for i, rec in enumerate(records):
            fr = rec["frame"]
            n = fr["pred_masks"].shape[2]
            pose = empty
            if i in kept:
                pose = dict(pred_RTs=rts[pos:pos + n], pred_scales=scales[pos:pos + n])
                pos += n
            if not alive[i]:
                continue
            det = {k: v for k, v in fr.items() if k not in ("pred_masks", "depth")}
            det.update(pose)
            det.update(rec.get("gts", {}))
            out.append(det)
        return out

    def run(self, dataset, camK=lde.REAL_INTRINSICS):
        """dataset: iterable of {'frame': {'depth','pred_masks','pred_bboxes','pred_class_ids','pred_scores'}, 'gts': {...}}
        (None entries are skipped, as the reference skips them :66-67) -> pred_results list."""
        results, chunk, pending = [], [], None
        was = getattr(self.net1, "graph_replay", False)
        self.net1.graph_replay = bool(self.graph) or was

        def flush():
            nonlocal pending, chunk
            launched = self._launch(chunk, camK)
            chunk = []
            if pending is not None:
                results.extend(self._finish(pending))
            pending = launched
            if not self.overlap:
                results.extend(self._finish(pending))
                pending = None
        for rec in dataset:
            if rec is None:
                continue
            chunk.append(rec)
            if len(chunk) == self.frames_per_batch:
                flush()
        if chunk:
            flush()
        if pending is not None:
            results.extend(self._finish(pending))
        self.net1.graph_replay = was
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
