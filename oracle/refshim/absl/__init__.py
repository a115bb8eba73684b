from . import flags, logging  # noqa: F401
