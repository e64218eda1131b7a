from . import prune  # noqa: F401
