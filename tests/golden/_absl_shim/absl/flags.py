"""absl.flags stand-in: DEFINE_* just records the default on a plain FLAGS namespace."""


class _Flags(object):
    def __call__(self, argv=None):
        return argv


FLAGS = _Flags()


def _define(name, default, help=None, **kw):
    setattr(FLAGS, name, default)


DEFINE_integer = DEFINE_string = DEFINE_float = DEFINE_bool = DEFINE_boolean = _define
DEFINE_list = DEFINE_enum = _define
