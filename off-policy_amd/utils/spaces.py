"""Minimal action/observation space descriptors.

The reference takes gym spaces or plain lists in `policy_info` (offpolicy/utils/rec_buffer.py:111-118,
offpolicy/utils/util.py:220-281). gym is not a dependency of the update path, so these duck-typed
stand-ins (same class names, same attributes) are accepted anywhere a gym space is.
"""
import numpy as np


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box(object):
    def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = tuple(shape) if shape is not None else np.asarray(low).shape


def get_dim_from_space(space):
    """Same contract as offpolicy/utils/util.py:220-237 for the space kinds the hot path uses."""
    name = space.__class__.__name__
    if name == "Box":
        return int(space.shape[0])
    if name == "Discrete":
        return int(space.n)
    if name == "list":
        return int(space[0])
    if isinstance(space, (int, np.integer)):
        return int(space)
    raise NotImplementedError("unsupported space %r" % (space,))
