"""Ring-slot bookkeeping of the episode replay (pure host logic, no device code).

Mirrors the index arithmetic of RecPolicyBuffer.insert (offpolicy/utils/rec_buffer.py:167-171, 187-188):
slots are handed out consecutively from `current_i`, wrapping to 0; `filled_i` saturates at capacity.
"""
import numpy as np


class RingIndex(object):
    def __init__(self, capacity):
        assert capacity > 0
        self.capacity = int(capacity)
        self.filled_i = 0
        self.current_i = 0

    def __len__(self):
        return self.filled_i

    def next_slots(self, n):
        """Slots for `n` new episodes (may wrap), and advance. Same result as the reference's idx_range."""
        n = int(n)
        assert 0 < n <= self.capacity, "cannot insert more episodes than the buffer holds"
        if self.current_i + n <= self.capacity:
            idx = np.arange(self.current_i, self.current_i + n)
        else:
            left = self.current_i + n - self.capacity
            idx = np.concatenate((np.arange(self.current_i, self.capacity), np.arange(left)))
        self.current_i = int(idx[-1]) + 1
        self.filled_i = min(self.filled_i + len(idx), self.capacity)
        return idx
