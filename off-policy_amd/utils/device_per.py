"""Device-resident sum/min segment trees for the prioritized replay buffers.

Host twin: utils/segment_tree.py (the reference's offpolicy/utils/segment_tree.py:18-165). With the trees in HBM a
prioritized update needs no host round trip: `sample` draws its B uniform masses on the host generator like the reference
(np.random.random), ships them through a pinned staging ring, and gets indices and importance weights back as device
tensors; `set` takes priorities straight from the trainer's device output."""
import ctypes as C

import numpy as np
import torch

from .. import _lib


class DevicePerTree(object):
    def __init__(self, capacity, alpha, device):
        assert capacity > 0 and capacity & (capacity - 1) == 0, "capacity must be positive and a power of 2."
        self.capacity, self.alpha, self.device = int(capacity), float(alpha), torch.device(device)
        nbytes = int(_lib.lib.ope_per_tree_bytes(self.capacity))
        self.trees = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        _lib.check(_lib.lib.ope_per_tree_init(_lib.ptr(self.trees), self.capacity, _lib.current_stream()), "ope_per_tree_init")
        self._ring = [(torch.empty(1024, dtype=torch.float64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        self._slot, self._used = 0, [False] * 8

    def _dev_idx(self, idx):
        if torch.is_tensor(idx):
            return idx.to(self.device, dtype=torch.int64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(idx, dtype=np.int64))).to(self.device)

    def set_to_max(self, idx):
        """New slots enter at the running maximum priority (rec_buffer.py:262-270 with the A-3 fix: every slot)."""
        idx = self._dev_idx(idx)
        _lib.check(_lib.lib.ope_per_tree_set(_lib.ptr(self.trees), self.capacity, _lib.ptr(idx), None, self.alpha, int(idx.numel()),
                                             _lib.current_stream()), "ope_per_tree_set")
        self._keep = idx

    def set(self, idx, priorities):
        idx = self._dev_idx(idx)
        if torch.is_tensor(priorities):
            pr = priorities.to(self.device, dtype=torch.float32).contiguous()
        else:
            pr = torch.from_numpy(np.ascontiguousarray(np.asarray(priorities, dtype=np.float32))).to(self.device)
        assert pr.numel() == idx.numel()
        _lib.check(_lib.lib.ope_per_tree_set(_lib.ptr(self.trees), self.capacity, _lib.ptr(idx), _lib.ptr(pr), self.alpha, int(idx.numel()),
                                             _lib.current_stream()), "ope_per_tree_set")
        self._keep = (idx, pr)

    def sample(self, mass01, filled, beta):
        """mass01: numpy float64 [B] in [0, 1). Returns (indices int64 [B], weights float32 [B]) on the device."""
        B = int(len(mass01))
        k = self._slot
        self._slot = (k + 1) % len(self._ring)
        host, ev = self._ring[k]
        if host.numel() < B:
            host = torch.empty(B, dtype=torch.float64).pin_memory()
            self._ring[k] = (host, ev)
        if self._used[k]:
            ev.synchronize()
        host[:B].copy_(torch.from_numpy(np.asarray(mass01, dtype=np.float64)))
        mass = torch.empty(B, dtype=torch.float64, device=self.device)
        mass.copy_(host[:B], non_blocking=True)
        ev.record()
        self._used[k] = True
        idx = torch.empty(B, dtype=torch.int64, device=self.device)
        w = torch.empty(B, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib.ope_per_tree_sample(_lib.ptr(self.trees), self.capacity, int(filled), _lib.ptr(mass), float(beta), B, _lib.ptr(idx),
                                                _lib.ptr(w), _lib.current_stream()), "ope_per_tree_sample")
        return idx, w

    def sample_dev(self, mass01, filled_dev, beta_dev):
        """`sample` with everything on the device: mass01 float64 [B] tensor (e.g. torch.rand inside a graph capture), the filled
        count (int32 [1], kept by the buffer) and beta (float64 [1], written by the caller before a replay). No host data, and a
        launch that is identical from call to call: capturable."""
        B = int(mass01.numel())
        assert mass01.dtype == torch.float64 and mass01.is_contiguous() and beta_dev.dtype == torch.float64 and filled_dev.dtype == torch.int32
        idx = torch.empty(B, dtype=torch.int64, device=self.device)
        w = torch.empty(B, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib.ope_per_tree_sample_dev(_lib.ptr(self.trees), self.capacity, _lib.ptr(filled_dev), _lib.ptr(mass01), _lib.ptr(beta_dev), B,
                                                    _lib.ptr(idx), _lib.ptr(w), _lib.current_stream()), "ope_per_tree_sample_dev")
        return idx, w

    # host views for tests / checkpoints
    def leaves(self):
        t = self.trees.view(torch.float64).cpu().numpy()
        c = self.capacity
        return t[c:2 * c].copy(), t[3 * c:4 * c].copy(), float(t[4 * c])

    def roots(self):
        t = self.trees.view(torch.float64)
        return float(t[1]), float(t[2 * self.capacity + 1])
