"""Hyper-parameters the point-cloud forward path reads.

The reference keeps these as absl flags in a global singleton (``config/config.py:7-150``) that the
modules read inside ``__init__``/``forward`` (``FaceRecon.py:15-17,54``, ``PoseR.py:13-14``,
``PoseTs.py:15-16``, ``PoseNet9D.py:69``).  Here the same names live on a plain object.  When the
host application already runs absl with the reference's ``config.config`` imported, ``FLAGS``
proxies every lookup to ``absl.flags.FLAGS`` so ``FLAGS.train`` set by ``evaluation.evaluate``
(``evaluation/evaluate.py:43``) is honoured; otherwise the reference defaults below apply.
"""

_DEFAULTS = dict(
    obj_c=6,             # config/config.py:7
    feat_c_R=1286,       # :36
    R_c=4,               # :37
    feat_c_ts=1289,      # :38
    Ts_c=6,              # :39
    gcn_sup_num=7,       # :44
    gcn_n_num=20,        # :45
    random_points=1024,  # :48
    train=1,             # :53
    output_channels=2500,  # :150
    # weights of the loss terms (losses/TDA_loss_sym_recon.py, losses/consistency_loss.py)
    fsnet_loss_type="l1",  # :68
    rot_1_w=8.0, rot_2_w=8.0,  # :71-72
    rot_regular=4.0,     # :74
    tran_w=8.0, size_w=8.0, recon_w=8.0,  # :75-77
    r_con_w=1.0,         # :78
    h1_w=4.0, h2_w=4.0,  # :79-80
    feat_consist_w=2.0,  # :83
    DCD_align=1.0,       # :101
    prop_sym_w=1.0,      # :114
    lr=1e-4, lr_pose=1.0,  # :118-120 (the trainer's parameter group)
)


class _Flags(object):
    def __init__(self):
        object.__setattr__(self, "_local", dict(_DEFAULTS))

    @staticmethod
    def _absl():
        try:
            import absl.flags as _af  # noqa: F401  (absent in the build image)
        except Exception:
            return None
        return _af.FLAGS

    def __getattr__(self, name):
        local = object.__getattribute__(self, "_local")
        ext = self._absl()
        if ext is not None:
            try:
                return getattr(ext, name)
            except Exception:
                pass
        if name in local:
            return local[name]
        raise AttributeError("unknown flag %r" % name)

    def __setattr__(self, name, value):
        ext = self._absl()
        if ext is not None:
            try:
                setattr(ext, name, value)
                return
            except Exception:
                pass
        object.__getattribute__(self, "_local")[name] = value


FLAGS = _Flags()
