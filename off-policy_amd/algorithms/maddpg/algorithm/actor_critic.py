"""MADDPG / MATD3 actor and centralised critic: parameter containers over flat CUDA vectors.

Mirrors offpolicy/algorithms/maddpg/algorithm/actor_critic.py:7-87. Both are an MLPBase trunk plus Linear head(s) and
share the 16-tensor MLP flat layout. Upstream defect A-4: the critic's `q_outs` is a plain Python list, so
the Q heads are NOT parameters -- never optimised, never soft-updated, absent from `state_dict()`, and the target
critic keeps its own random heads. `frozen_q_head=True` (default) reproduces exactly that: the heads live at the tail
of the flat vector but are not registered and the optimizer / Polyak only touch the prefix in front of them.
`frozen_q_head=False` registers them (`q_outs.{k}.weight/bias`) and trains them.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .... import _lib
from ....flat_module import FlatModule

H = 64
_TRUNK = ["mlp.feature_norm.weight", "mlp.feature_norm.bias",
          "mlp.mlp.fc1.0.weight", "mlp.mlp.fc1.0.bias", "mlp.mlp.fc1.2.weight", "mlp.mlp.fc1.2.bias",
          "mlp.mlp.fc_h.0.weight", "mlp.mlp.fc_h.0.bias", "mlp.mlp.fc_h.2.weight", "mlp.mlp.fc_h.2.bias",
          "mlp.mlp.fc2.0.0.weight", "mlp.mlp.fc2.0.0.bias", "mlp.mlp.fc2.0.2.weight", "mlp.mlp.fc2.0.2.bias"]


def _trunk_shapes(D):
    return [(D,), (D,), (H, D), (H,), (H,), (H,), (H, H), (H,), (H,), (H,), (H, H), (H,), (H,), (H,)]


def ddpg_layout(cfg, which):
    off, siz = (C.c_int64 * 16)(), (C.c_int64 * 16)()
    total = _lib.lib.ope_ddpg_param_layout(C.byref(cfg), which, off, siz)
    if total < 0:
        _lib.check(int(total), "ope_ddpg_param_layout")
    return list(off), list(siz), int(total)


def _draw_trunk(in_dim, use_orthogonal, use_ReLU):
    """MLPBase construction draws (mlp.py:52-74): fc1 then fc_h (fc2[0] is its deep copy)."""
    init_w = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
    gain = nn.init.calculate_gain("relu" if use_ReLU else "tanh")
    fc1 = nn.Linear(in_dim, H)
    init_w(fc1.weight.data, gain=gain)
    fch = nn.Linear(H, H)
    init_w(fch.weight.data, gain=gain)
    one, zero = torch.ones, torch.zeros
    return [one(in_dim), zero(in_dim), fc1.weight.data, zero(H), one(H), zero(H), fch.weight.data, zero(H), one(H), zero(H),
            fch.weight.data.clone(), zero(H), one(H), zero(H)]


def draw_actor_values(args, obs_dim, act_dim):
    """`act_dim`: an int, or the array of a multi-discrete space's sub-action sizes -- one Linear head each, constructed in order
    (ACTLayer, act.py:14-19); the heads' rows are returned stacked ([sum, 64] / [sum]: the flat layout of a single head)."""
    vals = _draw_trunk(obs_dim, getattr(args, "use_orthogonal", True), getattr(args, "use_ReLU", True))
    init_w = nn.init.orthogonal_ if getattr(args, "use_orthogonal", True) else nn.init.xavier_uniform_
    ws = []
    for a_dim in ([int(act_dim)] if np.ndim(act_dim) == 0 else [int(x) for x in act_dim]):
        out = nn.Linear(H, a_dim)
        init_w(out.weight.data, gain=getattr(args, "gain", 0.01))          # ACTLayer, act.py:10-19
        ws.append(out.weight.data)
    w = torch.cat(ws, dim=0)
    return [v.detach().float() for v in vals + [w, torch.zeros(w.shape[0])]]


def draw_critic_values(args, in_dim, num_q):
    vals = _draw_trunk(in_dim, getattr(args, "use_orthogonal", True), getattr(args, "use_ReLU", True))
    init_w = nn.init.orthogonal_ if getattr(args, "use_orthogonal", True) else nn.init.xavier_uniform_
    ws = []
    for _ in range(num_q):                                               # actor_critic.py:64-67, gain 1
        q = nn.Linear(H, 1)
        init_w(q.weight.data)
        ws.append(q.weight.data.reshape(H))
    return [v.detach().float() for v in vals + [torch.stack(ws), torch.zeros(num_q)]]


class _Head(object):
    """Unregistered Linear(64, 1) head (what upstream's plain-list `q_outs` entries amount to)."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


class MADDPG_Actor(FlatModule):
    def __init__(self, args, obs_dim, act_dim, device, cfg, flat=None, values=None):
        offs, sizes, total = ddpg_layout(cfg, 0)
        device = torch.device(device)
        if flat is None:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
        heads = None if np.ndim(act_dim) == 0 else [int(x) for x in act_dim]
        if heads is None:
            names = _TRUNK + ["act.action_out.weight", "act.action_out.bias"]
            shapes, o = _trunk_shapes(obs_dim) + [(int(act_dim), H), (int(act_dim),)], offs
        else:
            # multi-discrete: act.action_outs.{i}.{weight,bias} (act.py:14-17) are row ranges of the one stacked head the kernels see
            names, shapes, o, r0 = list(_TRUNK), _trunk_shapes(obs_dim), list(offs[:14]), 0
            for i, a_dim in enumerate(heads):
                names += ["act.action_outs.%d.weight" % i, "act.action_outs.%d.bias" % i]
                shapes += [(a_dim, H), (a_dim,)]
                o += [offs[14] + r0 * H, offs[15] + r0]
                r0 += a_dim
            act_dim = r0
        super().__init__(names, shapes, o, flat)
        self.head_dims = heads
        self.obs_dim, self.act_dim, self.device, self.padded_numel = int(obs_dim), int(act_dim), device, total
        self._dims = _lib.Dims(1, int(act_dim), int(obs_dim), 1, 1)
        self._ws = None
        if values is not None:
            for p, v in zip(list(self.parameters())[:14], values[:14]):
                p.data.copy_(v)
            A = self.act_dim
            flat[offs[14]:offs[14] + A * H].view(A, H).copy_(values[14])
            flat[offs[15]:offs[15] + A].copy_(values[15])

    def forward(self, x):
        """Logits for every action (actor_critic.py:28-41) through ope_agent_forward_mlp."""
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        rows = int(x.shape[0])
        need = _lib.lib.ope_agent_forward_mlp_workspace_bytes(C.byref(self._dims), rows)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        out = torch.empty((rows, self.act_dim), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib.ope_agent_forward_mlp(C.byref(self._dims), rows, _lib.ptr(x), _lib.ptr(self._flat), _lib.ptr(self._ws),
                                                  self._ws.numel(), _lib.ptr(out), _lib.current_stream()), "ope_agent_forward_mlp")
        return out

    __call__ = forward


class MADDPG_Critic(FlatModule):
    def __init__(self, args, central_obs_dim, central_act_dim, device, cfg, num_q_outs=1, flat=None, values=None,
                 frozen_q_head=True):
        offs, sizes, total = ddpg_layout(cfg, 1)
        device = torch.device(device)
        if flat is None:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
        in_dim = int(central_obs_dim + central_act_dim)
        names, shapes, o = list(_TRUNK), _trunk_shapes(in_dim), offs[:14]
        if not frozen_q_head:
            names += ["q_outs.weight", "q_outs.bias"]
            shapes += [(num_q_outs, H), (num_q_outs,)]
            o = offs
        super().__init__(names, shapes, o, flat)
        self.input_dim, self.num_q_outs, self.device, self.padded_numel = in_dim, int(num_q_outs), device, total
        self.frozen_q_head = bool(frozen_q_head)
        self.head_offset = int(offs[14])                 # everything before this is what Adam / Polyak touch when frozen
        hw = flat[offs[14]:offs[14] + num_q_outs * H].view(num_q_outs, H)
        hb = flat[offs[15]:offs[15] + num_q_outs]
        if self.frozen_q_head:   # upstream's plain (unregistered) list; in fixed mode `q_outs` is the registered sub-module
            self.q_outs = [_Head(hw[k], hb[k:k + 1]) for k in range(num_q_outs)]
        self._head_w, self._head_b = hw, hb
        if values is not None:
            for p, v in zip(list(self.parameters())[:14], values[:14]):
                p.data.copy_(v)
            hw.copy_(values[14])
            hb.copy_(values[15])

    @property
    def trainable_numel(self):
        return self.head_offset if self.frozen_q_head else self.padded_numel
