"""Pose assembly for the evaluation driver.

``generate_RT`` stands where ``tools.geom_utils.generate_RT`` stands in the reference
(``evaluater/RT_TDA_Evaluater.py:7,94``); the reference ships that module only as py3.8 bytecode, its semantics
are recorded in SURVEY.md section 8c.  ``batched_inference`` is the cross-image batching the reference lacks: it
runs one forward over the detections of many images and returns per-image ``pred_RTs`` / ``pred_scales`` with a
single device-to-host copy at the end (the reference synchronises per image, RT_TDA_Evaluater.py:97-98).
"""
import torch

from . import ops


def generate_RT(R, f, T, mode="vec", sym=None):
    """R = [p_green (B,3), p_red (B,3)], f = [f_green (B,), f_red (B,)], T (B,3), sym (B,4) -> (B,4,4)"""
    if mode != "vec":
        raise NotImplementedError("only mode='vec' is used by the reference's evaluater")
    return ops.generate_rt(R[0], R[1], f[0], f[1], T, sym)


def infer_device(net, pts, cat, ms, sym, max_batch=256, eval_outputs_only=None):
    """One or more forwards over (n,N,3) clouds already on the device -> (pred_RTs (n,4,4), pred_scales (n,3)) device tensors;
    nothing synchronises (the caller decides when to copy back).  eval_outputs_only: handed to each forward (None = the net's /
    the process's setting): the six pose outputs are all this function reads."""
    rts, scales = [], []
    kw = {} if eval_outputs_only is None else dict(eval_outputs_only=bool(eval_outputs_only))
    for lo in range(0, pts.shape[0], max_batch):
        out = net(pts[lo:lo + max_batch], cat[lo:lo + max_batch], **kw)
        rts.append(generate_RT([out["p_green_R"], out["p_red_R"]], [out["f_green_R"], out["f_red_R"]], out["Pred_T"],
                               mode="vec", sym=sym[lo:lo + max_batch]))
        scales.append(out["Pred_s"] + ms[lo:lo + max_batch])
    return torch.cat(rts), torch.cat(scales)


def batched_inference(net, clouds, cat_ids, mean_shapes, syms, max_batch=256):
    """clouds: list over images of (n_det_i, N, 3) tensors (same N); cat_ids / mean_shapes / syms likewise.
    Returns a list over images of dicts {'pred_RTs': (n_det_i,4,4) ndarray, 'pred_scales': (n_det_i,3) ndarray}."""
    counts = [c.shape[0] for c in clouds]
    keep = [i for i, n in enumerate(counts) if n > 0]
    results = [dict(pred_RTs=torch.zeros(0, 4, 4).numpy(), pred_scales=torch.zeros(0, 4, 4).numpy()) for _ in counts]
    if not keep:
        return results
    dev = next(net.parameters()).device
    pts = torch.cat([clouds[i] for i in keep]).to(dev).float()
    cat = torch.cat([cat_ids[i].reshape(-1, 1) for i in keep]).to(dev).float()
    ms = torch.cat([mean_shapes[i] for i in keep]).to(dev).float()
    sym = torch.cat([syms[i] for i in keep]).to(dev).float()
    rts, scales = infer_device(net, pts, cat, ms, sym, max_batch)
    rts, scales = rts.cpu().numpy(), scales.cpu().numpy()                           # the only device-to-host copies
    pos = 0
    for i in keep:
        results[i] = dict(pred_RTs=rts[pos:pos + counts[i]], pred_scales=scales[pos:pos + counts[i]])
        pos += counts[i]
    return results
