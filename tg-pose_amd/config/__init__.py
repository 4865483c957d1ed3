from .flags import FLAGS  # noqa: F401
