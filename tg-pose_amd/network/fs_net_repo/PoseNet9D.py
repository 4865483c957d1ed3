"""Drop-in for ``network/fs_net_repo/PoseNet9D.py`` (module seam #1 of SURVEY.md section 8b).

``PoseNet9D(only_encoder=False).forward(points, obj_id, enable_proj=False) -> dict`` with the
reference's key sets (PoseNet9D.py:69-90: 11 keys when FLAGS.train, 6 otherwise), submodule
names (face_all / face_enc, rot_green, rot_red, ts) and therefore state-dict keys, so
``load_state_dict(checkpoint['net1_state_dict'])`` (evaluater/RT_TDA_Evaluater.py:39) works
unchanged.  The forward is the eval-mode HIP pipeline; extra keyword-only arguments let tests
pin the random subsample and inject / record neighbour graphs.
"""
import torch
import torch.nn as nn

from ... import engine
from ...config import FLAGS
from .FaceRecon import FaceNet, _WithBuffers
from .gcn3d import _need_eval
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
        face = "face_enc." if self.only_encoder else "face_all."
        return self._packed(lambda: engine.Packed(self.state_dict(), device, face=face,
                                                  with_heads=not self.only_encoder))

    def forward(self, points, obj_id, enable_proj=False, *, sample_idx=None, inject=None, record=None):
        _need_eval(self)
        if enable_proj:
            raise NotImplementedError("enable_proj=True is never used by the reference's train/eval path")
        if not points.is_cuda:
            raise RuntimeError("tgpose_amd.PoseNet9D runs on the GPU only (no CPU fallback); move inputs to cuda")
        pk = self.packed(points.device)
        with torch.no_grad():
            if self.only_encoder:
                return engine.encoder_only_forward(pk, points, obj_id, sample_idx, inject, record,
                                                   FLAGS.gcn_n_num, FLAGS.obj_c)
            return engine.posenet_forward(pk, points, obj_id, bool(FLAGS.train), sample_idx, inject, record,
                                          FLAGS.gcn_n_num, FLAGS.obj_c)
