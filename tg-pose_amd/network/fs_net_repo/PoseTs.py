"""Drop-in for ``network/fs_net_repo/PoseTs.py``: translation + size head (reference PoseTs.py:12-45)."""
from ...config import FLAGS
from .PoseR import _PointHead


class Pose_Ts(_PointHead):
    def __init__(self):
        super().__init__(FLAGS.feat_c_ts, FLAGS.Ts_c)

    def forward(self, x):
        out = self._run(x)
        return out[:, 0:3], out[:, 3:6]
