"""Drop-in for ``losses/consistency_loss.py``: ``feat_consistency_loss`` (:11-16) and ``prop_sym_matching_loss`` (:19-48),
each one forward launch pair and one backward launch on the HIP kernels of csrc/tdaloss.hip, differentiable with respect to
every tensor the reference's versions are (both feature sets; both clouds -- the trainer passes a reconstruction as ``PC``
at trainer/RL_TDA.py:132)."""
import torch
from torch.autograd import Function

from .. import ops
from ..config.flags import FLAGS


class _FeatConsistency(Function):
    @staticmethod
    def forward(ctx, x1, x2):
        loss, rows = ops.feat_consistency_fwd(x1, x2)
        ctx.save_for_backward(x1, x2, rows)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x1, x2, rows = ctx.saved_tensors
        d1, d2 = ops.feat_consistency_bwd(x1, x2, rows, g.reshape(1).float().contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return d1, d2


class _SymRecon(Function):
    @staticmethod
    def forward(ctx, PC, PC_re, gt_R, gt_t, sym):
        ctx.save_for_backward(PC, PC_re, gt_R, gt_t, sym)
        return ops.sym_recon_fwd(PC, PC_re, gt_R, gt_t, sym).reshape(())

    @staticmethod
    def backward(ctx, g):
        PC, PC_re, gt_R, gt_t, sym = ctx.saved_tensors
        need_pc, need_re = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_pc or need_re):
            return None, None, None, None, None
        dPC, dRe = ops.sym_recon_bwd(PC, PC_re, gt_R, gt_t, sym, g.reshape(1).float().contiguous(), need_pc, need_re)
        return dPC, dRe, None, None, None


def _f(t):
    return t.float().contiguous()


def feat_consistency(x1, x2):
    """2 - 2 sum_b <x1_b/|x1_b|, x2_b/|x2_b|> / B   (TDA_loss.feat_consist_loss :113-117, unweighted)"""
    return _FeatConsistency.apply(_f(x1), _f(x2))


def feat_consistency_loss(x1, x2):
    return FLAGS.feat_consist_w * feat_consistency(x1, x2)


def prop_sym_matching_loss(PC, PC_re, gt_R, gt_t, sym):
    """L1 between the reconstruction PC_re and PC mapped by each object's own symmetry in the ground-truth frame."""
    return _SymRecon.apply(_f(PC), _f(PC_re), _f(gt_R), _f(gt_t), ops.sym_i32(sym))
