"""nn.Module façade over a flat CUDA parameter vector.

The kernels read ONE flat float32 vector per network set (layout from `ope_qmix_param_layout`). The reference's
runners, however, reach into `policy.q_network` / `trainer.mixer` as nn.Modules: `state_dict()` for checkpoints
(offpolicy/runner/rnn/base_runner.py:303-315), `parameters()` for the optimizer list, `.train()/.eval()`.
`FlatModule` gives them that surface: every parameter is an nn.Parameter whose storage IS a slice of the flat
vector, registered under the reference's dotted names, so `named_parameters()` / `state_dict()` keys and order
match the reference and rollout code always sees the live weights.
"""
from collections import OrderedDict

import torch
import torch.nn as nn


class _Node(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("FlatModule parameters are consumed by ope kernels, not by nn.Module.forward")


class FlatModule(_Node):
    def __init__(self, names, shapes, offsets, flat):
        """names/shapes/offsets: parallel lists (offsets in floats into `flat`)."""
        super().__init__()
        self._spec = list(zip(names, [tuple(s) for s in shapes], [int(o) for o in offsets]))
        self._flat = None
        for name, shape, off in self._spec:
            mod = self
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Node())
                mod = mod._modules[p]
            mod.register_parameter(parts[-1], nn.Parameter(torch.empty(0), requires_grad=False))
        self.rebind(flat)

    def rebind(self, flat):
        """Point every parameter at its slice of `flat` (values are whatever `flat` holds)."""
        self._flat = flat
        params = dict(self.named_parameters())
        for name, shape, off in self._spec:
            n = 1
            for s in shape:
                n *= s
            params[name].data = flat[off:off + n].view(shape)

    def copy_values_from(self, other):
        for (k, p), (k2, q) in zip(self.named_parameters(), other.named_parameters()):
            assert k == k2
            p.data.copy_(q.data)

    def spec(self):
        return OrderedDict((n, (s, o)) for n, s, o in self._spec)

    @property
    def unused_range(self):
        """[begin, end) of the registered-but-unused `fc_h` block in the flat vector (mlp.py:21-23): tensors that never get a
        gradient, which torch's Adam therefore never touches (no weight decay either). (0, 0) if the module has none."""
        cached = self.__dict__.get("_unused_range")
        if cached is not None:
            return cached
        spans = []
        for name, shape, off in self._spec:
            if ".fc_h." in name:
                n = 1
                for d in shape:
                    n *= d
                spans.append((off, off + n))
        r = (min(a for a, _ in spans), max(b for _, b in spans)) if spans else (0, 0)
        self.__dict__["_unused_range"] = r
        return r
