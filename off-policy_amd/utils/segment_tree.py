"""Sum / min segment trees for proportional prioritized replay (host side, float64, numpy).

Same contract as the reference's offpolicy/utils/segment_tree.py:18-165 (vectorised `__setitem__`, `reduce`,
`find_prefixsum_idx`), written as an iterative bottom-up array tree: leaves live at [capacity, 2*capacity),
node i has children 2i and 2i+1. PER bookkeeping is latency-trivial (~0.1 ms) and stays
on the host; a device tree is listed as "next" (section 8(f) 4).
"""
import numpy as np


class SegmentTree(object):
    def __init__(self, capacity, op, neutral):
        assert capacity > 0 and capacity & (capacity - 1) == 0, "capacity must be positive and a power of 2."
        self._capacity = capacity
        self._op = op
        self._neutral = neutral
        self._value = np.full(2 * capacity, neutral, dtype=np.float64)

    def reduce(self, start=0, end=None):
        """op over the half-open leaf range [start, end) (reference semantics: `end` exclusive after its `-= 1`)."""
        if end is None:
            end = self._capacity
        if end < 0:
            end += self._capacity
        res = self._neutral
        lo, hi = start + self._capacity, end + self._capacity
        while lo < hi:
            if lo & 1:
                res = self._op(res, self._value[lo])
                lo += 1
            if hi & 1:
                hi -= 1
                res = self._op(res, self._value[hi])
            lo >>= 1
            hi >>= 1
        return res

    def __setitem__(self, idx, val):
        idx = np.atleast_1d(np.asarray(idx, dtype=np.int64))
        val = np.broadcast_to(np.asarray(val, dtype=np.float64), idx.shape)
        assert np.all((idx >= 0) & (idx < self._capacity))
        pos = idx + self._capacity
        self._value[pos] = val          # duplicate indices: last write wins, as numpy fancy assignment does
        pos = np.unique(pos >> 1)       # all touched nodes sit on one level: walk up level by level
        while pos[0] >= 1:
            self._value[pos] = self._op(self._value[2 * pos], self._value[2 * pos + 1])
            pos = np.unique(pos >> 1)

    def __getitem__(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        assert np.all((idx >= 0) & (idx < self._capacity))
        return self._value[idx + self._capacity]


class SumSegmentTree(SegmentTree):
    def __init__(self, capacity):
        super().__init__(capacity, np.add, 0.0)

    def sum(self, start=0, end=None):
        return self.reduce(start, end)

    def find_prefixsum_idx(self, prefixsum):
        """Highest i with sum(leaf[0..i-1]) <= prefixsum, vectorised over an array of prefix sums
        (reference: segment_tree.py:115-146)."""
        scalar = np.isscalar(prefixsum)
        p = np.atleast_1d(np.asarray(prefixsum, dtype=np.float64)).copy()
        assert np.all(p >= 0) and np.all(p <= self.sum() + 1e-5)
        idx = np.ones(p.shape, dtype=np.int64)
        while idx[0] < self._capacity:          # all lanes descend one level per iteration
            left = 2 * idx
            lv = self._value[left]
            go_right = lv <= p
            p = np.where(go_right, p - lv, p)
            idx = np.where(go_right, left + 1, left)
        out = idx - self._capacity
        return int(out[0]) if scalar else out


class MinSegmentTree(SegmentTree):
    def __init__(self, capacity):
        super().__init__(capacity, np.minimum, float("inf"))

    def min(self, start=0, end=None):
        return self.reduce(start, end)
