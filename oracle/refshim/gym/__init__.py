"""Stub of the `gym` package: just enough surface for the reference's hot-path modules to import.

TEST INFRASTRUCTURE ONLY (see oracle/README.md). The reference (marlbenchmark/off-policy) imports gym at
offpolicy/utils/util.py:2,4 for space classes; the update path never calls into gym itself.
"""
from . import spaces  # noqa: F401


class Space(object):
    pass


class Env(object):
    pass
