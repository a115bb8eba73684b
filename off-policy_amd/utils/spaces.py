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


class MultiDiscrete(object):
    """offpolicy/utils/util.py:35-73: a list of [min, max] pairs, one discrete sub-action each (e.g. MPE agents that move AND communicate,
    environment.py:66-75). `high - low + 1` are the sub-actions' sizes; an action is the concatenation of their one-hot blocks."""

    def __init__(self, array_of_param_array):
        self.low = np.array([x[0] for x in array_of_param_array])
        self.high = np.array([x[1] for x in array_of_param_array])
        self.num_discrete_space = self.low.shape[0]
        self.n = int(np.sum(self.high) + 2)
        self.shape = (self.num_discrete_space,)

    def __repr__(self):
        return "MultiDiscrete" + str(self.num_discrete_space)


def get_dim_from_space(space):
    """Same contract as offpolicy/utils/util.py:220-237 for the space kinds the hot path uses (a MultiDiscrete space gives the ARRAY of
    its sub-actions' sizes, as upstream)."""
    name = space.__class__.__name__
    if "MultiDiscrete" in name:
        return (np.asarray(space.high) - np.asarray(space.low)) + 1
    if name == "Box":
        return int(space.shape[0])
    if name == "Discrete":
        return int(space.n)
    if name == "list":
        return int(space[0])
    if isinstance(space, (int, np.integer)):
        return int(space)
    raise NotImplementedError("unsupported space %r" % (space,))
