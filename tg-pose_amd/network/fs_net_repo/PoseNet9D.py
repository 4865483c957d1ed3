"""Drop-in for ``network/fs_net_repo/PoseNet9D.py`` (module seam #1 of SURVEY.md section 8b).

``PoseNet9D(only_encoder=False).forward(points, obj_id, enable_proj=False) -> dict`` with the
reference's key sets (PoseNet9D.py:69-90: 11 keys when FLAGS.train, 6 otherwise), submodule
names (face_all / face_enc, rot_green, rot_red, ts) and therefore state-dict keys, so
``load_state_dict(checkpoint['net1_state_dict'])`` (evaluater/RT_TDA_Evaluater.py:39) works
unchanged.  The forward is the fused HIP pipeline of ``tgpose_amd.engine`` in eval mode and in training
mode under ``no_grad`` (batch-statistics BatchNorm, dropout: the trainer's net2), and the differentiable
one of ``tgpose_amd.autograd`` in training mode with gradients recorded (net1); extra keyword-only
arguments let tests pin the random subsample and inject / record neighbour graphs.
"""
import torch
import torch.nn as nn

from ... import engine, ops
from ...config import FLAGS
from .FaceRecon import FaceNet, _WithBuffers
from .PoseR import Rot_green, Rot_red
from .PoseTs import Pose_Ts


class PoseNet9D(_WithBuffers):
    def __init__(self, only_encoder=False):
        super().__init__()
        self.only_encoder = only_encoder
        if not only_encoder:
            self.face_all = FaceNet()
            self.rot_green = Rot_green()
            self.rot_red = Rot_red()
            self.ts = Pose_Ts()
        else:
            self.face_enc = FaceNet()

    def packed(self, device):
        """Kernel-ready weights: rebuilt when a parameter changes, BatchNorm folds refreshed when only the running
        statistics changed (a training-mode forward moves them)."""
        psig = (ops.GEMM_MODE,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        bsig = tuple((b.data_ptr(), b._version) for b in self.buffers())
        if getattr(self, "_pk_psig", None) != psig:
            face = "face_enc." if self.only_encoder else "face_all."
            self._pk = engine.Packed(self.state_dict(), device, face=face, with_heads=not self.only_encoder)
            self._pk_psig, self._pk_bsig = psig, bsig
        elif self._pk_bsig != bsig and not self.training:    # training mode never reads the folds: refresh them on the next eval call
            self._pk.refold(self.state_dict())
            self._pk_bsig = tuple((b.data_ptr(), b._version) for b in self.buffers())
        return self._pk

    def _proj_operands(self, device):
        """GEMM operands of Face_Enc.proj_layer, built on first use and rebuilt when its weights change"""
        enc = (self.face_enc if self.only_encoder else self.face_all).encoder
        sig = tuple((p.data_ptr(), p._version) for p in (enc.proj_layer[0].weight, enc.proj_layer[3].weight)) + (str(device),)
        if getattr(self, "_proj_sig", None) != sig:
            face = "face_enc." if self.only_encoder else "face_all."
            self._proj, self._proj_sig = engine.proj_operands(self.state_dict(), face, device), sig
        return self._proj

    def forward(self, points, obj_id, enable_proj=False, *, sample_idx=None, inject=None, record=None, cut=None,
                eval_outputs_only=None):
        if not points.is_cuda:
            raise RuntimeError("tgpose_amd.PoseNet9D runs on the GPU only (no CPU fallback); move inputs to cuda")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # differentiable path: the same kernels composed as torch.autograd.Functions (tgpose_amd/autograd.py)
            from ... import autograd as tgp_autograd
            return tgp_autograd.posenet_forward(self, points, obj_id, bool(FLAGS.train), sample_idx, inject, record,
                                                FLAGS.gcn_n_num, FLAGS.obj_c, cut=cut, enable_proj=bool(enable_proj))
        pk = self.packed(points.device)
        # enable_proj (PoseNet9D.py:33,39,49): feat_global goes through Face_Enc.proj_layer before its max over points.  It only
        # shows in the outputs that carry feat_global: the encoder-only net's, and the full net's with FLAGS.train set.
        proj = None
        if enable_proj and (self.only_encoder or bool(FLAGS.train)):
            proj = (self._proj_operands(points.device), self.state_dict())
        with torch.no_grad():
            if self.training:
                sd = self.state_dict()
                if self.only_encoder:
                    return engine.encoder_only_forward_train(pk, sd, points, obj_id, sample_idx, inject, record,
                                                             FLAGS.gcn_n_num, FLAGS.obj_c, proj=proj)
                p_ph = self.face_all.ph_pred.dp1.p
                p_hd = self.rot_green.drop1.p
                return engine.posenet_forward_train(pk, sd, points, obj_id, bool(FLAGS.train), sample_idx, inject, record,
                                                    FLAGS.gcn_n_num, FLAGS.obj_c, dropout_p=(p_ph, p_hd), proj=proj)
            if self.only_encoder:
                return engine.encoder_only_forward(pk, points, obj_id, sample_idx, inject, record,
                                                   FLAGS.gcn_n_num, FLAGS.obj_c, proj=proj)
            # eval outputs only (a deployment switch): the layers whose results the six-key eval dict does not return -- PH
            # predictor, decoder -- are not computed.
            # Per call (eval_outputs_only=..., what the evaluation driver passes), else the module's attribute, else the process
            # default; handed down as an argument -- nothing global is rewritten, so nets with different settings do not interfere.
            lean = eval_outputs_only if eval_outputs_only is not None else getattr(self, "eval_outputs_only", None)
            lean = engine.EVAL_OUTPUTS_ONLY if lean is None else bool(lean)
            if getattr(self, "graph_replay", False) and inject is None and record is None and proj is None:
                # opt-in (net.graph_replay = True): the forward of this (batch, cloud size, output set) is captured once as
                # a hipGraph and replayed; the returned tensors are the graph's static outputs, valid until the next call
                key = (id(pk), tuple(points.shape), bool(FLAGS.train), ops.GEMM_MODE, engine.BRANCH_STREAMS, lean)
                graphs = self.__dict__.setdefault("_graphs", {})
                if key not in graphs:
                    graphs.clear()                      # one resident graph: its private pool holds every activation
                    graphs[key] = engine.GraphedForward(pk, points.shape[0], points.shape[1], points.device, bool(FLAGS.train),
                                                        1, FLAGS.gcn_n_num, FLAGS.obj_c, outputs_only=lean)
                return dict(graphs[key](points, obj_id, sample_idx))
            return engine.posenet_forward(pk, points, obj_id, bool(FLAGS.train), sample_idx, inject, record,
                                          FLAGS.gcn_n_num, FLAGS.obj_c, outputs_only=lean, proj=proj)
