"""Space stubs. Class NAMES matter: the reference tests `obs_space.__class__.__name__ == 'Box'`
(offpolicy/utils/rec_buffer.py:111) and uses isinstance on Box/Discrete/Tuple (offpolicy/utils/util.py:220-281)."""
import numpy as np


class Box(object):
    def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = tuple(shape) if shape is not None else np.asarray(low).shape


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()


class Tuple(object):
    def __init__(self, spaces):
        self.spaces = list(spaces)

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]


class MultiDiscrete(object):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec)
        self.shape = self.nvec.shape
