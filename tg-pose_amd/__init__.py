"""tg-pose_amd: MI355X-native (gfx950) point-cloud forward path of TG-Pose.

    from tgpose_amd import PoseNet9D, chamfer_3DDist, FLAGS

mirrors ``network.fs_net_repo.PoseNet9D.PoseNet9D`` and ``losses.chamfer3D.dist_chamfer_3D`` of the
reference; the operator seam lives in ``tgpose_amd.network.fs_net_repo.gcn3d``.  See DESIGN.md.
"""
__version__ = "0.1.0"

from .config import FLAGS  # noqa: F401
from .init_weights import seeded_state_dict, state_spec  # noqa: F401


def __getattr__(name):
    # heavy imports (torch.nn modules) on first use
    if name == "PoseNet9D":
        from .network.fs_net_repo.PoseNet9D import PoseNet9D
        return PoseNet9D
    if name in ("chamfer_3DDist", "chamfer_3DFunction", "chamfer_3D"):
        from .losses.chamfer3D import dist_chamfer_3D
        return getattr(dist_chamfer_3D, name)
    raise AttributeError(name)
