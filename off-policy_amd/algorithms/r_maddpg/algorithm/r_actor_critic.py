"""Recurrent MADDPG / MATD3 actor and centralised critic: parameter containers over flat CUDA vectors.

Mirrors offpolicy/algorithms/r_maddpg/algorithm/r_actor_critic.py:7-129. Both are an RNNBase body (feature LayerNorm ->
fc1 -> fc2 -> GRU -> LayerNorm, offpolicy/algorithms/utils/rnn.py:28-47) plus a Linear head, i.e. the 22-tensor
recurrent layout of the QMIX agent network. Unlike the MLP family, the recurrent critic keeps its heads
in an nn.ModuleList, so `q_outs.{k}.weight/bias` are registered, trained and soft-updated.
`prev_act_inp` (feeding the previous action to the actor) is off in every reference config and is not supported.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .... import _lib
from ....flat_module import FlatModule
from ...qmix.algorithm.agent_q_function import AGENT_PARAM_NAMES, agent_param_shapes, init_agent_values

H = 64
_BODY = AGENT_PARAM_NAMES[:20]


def rddpg_layout(cfg, which):
    off, siz = (C.c_int64 * 22)(), (C.c_int64 * 22)()
    total = _lib.lib.ope_rddpg_param_layout(C.byref(cfg), which, off, siz)
    if total < 0:
        _lib.check(int(total), "ope_rddpg_param_layout")
    return list(off), list(siz), int(total)


def draw_ractor_values(args, obs_dim, act_dim):
    """RNNBase draws then ACTLayer(gain=args.gain) (r_actor_critic.py:28-31): the QMIX agent network's order. `act_dim` may be the array
    of a multi-discrete space's sub-action sizes: one Linear head each, constructed in order (act.py:14-17), returned stacked."""
    orth, gain = getattr(args, "use_orthogonal", True), getattr(args, "gain", 0.01)
    if np.ndim(act_dim) == 0:
        return init_agent_values(obs_dim, int(act_dim), orth, gain, getattr(args, "use_ReLU", True))
    heads = [int(x) for x in act_dim]
    vals = init_agent_values(obs_dim, heads[0], orth, gain, getattr(args, "use_ReLU", True))       # body + head 0
    init_w = nn.init.orthogonal_ if orth else nn.init.xavier_uniform_
    ws = [vals[20]]
    for a_dim in heads[1:]:
        out = nn.Linear(H, a_dim)
        init_w(out.weight.data, gain=gain)
        ws.append(out.weight.data.detach().float())
    w = torch.cat(ws, dim=0)
    return vals[:20] + [w, torch.zeros(w.shape[0])]


def draw_rcritic_values(args, in_dim, num_q):
    """RNNBase draws, then one Linear(64, 1) per Q head, gain 1 (r_actor_critic.py:84-89).
    Returns the 20 body tensors + [head weights [num_q, 64], head biases [num_q]]."""
    orth = getattr(args, "use_orthogonal", True)
    vals = init_agent_values(in_dim, 1, orth, 1.0, getattr(args, "use_ReLU", True))     # body + head 0
    init_w = nn.init.orthogonal_ if orth else nn.init.xavier_uniform_
    ws = [vals[20].reshape(H)]
    for _ in range(1, num_q):
        q = nn.Linear(H, 1)
        init_w(q.weight.data)
        ws.append(q.weight.data.reshape(H).detach().float())
    return vals[:20] + [torch.stack(ws), torch.zeros(num_q)]


class _RnnNet(FlatModule):
    """Shared forward: sequences or single steps through ope_agent_forward (RNNBase + head)."""

    def _setup_forward(self, in_dim, out_dim):
        self._dims = _lib.Dims(1, int(out_dim), int(in_dim), 1, 1)
        self._fws = None

    def _run(self, x, rnn_states):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        rnn_states = torch.as_tensor(rnn_states, dtype=torch.float32, device=self.device)
        no_sequence = x.dim() == 2
        if no_sequence:
            x = x[None]
        if rnn_states.dim() == 3:
            rnn_states = rnn_states[0]
        L, R = int(x.shape[0]), int(x.shape[1])
        x, h0 = x.contiguous(), rnn_states.contiguous()
        need = _lib.lib.ope_agent_forward_workspace_bytes(C.byref(self._dims), L, R)
        if self._fws is None or self._fws.numel() < need:
            self._fws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        out = torch.empty((L, R, self._dims.act_dim), dtype=torch.float32, device=self.device)
        h = torch.empty((L, R, H), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib.ope_agent_forward(C.byref(self._dims), L, R, _lib.ptr(x), _lib.ptr(h0), _lib.ptr(self._flat),
                                              _lib.ptr(self._fws), self._fws.numel(), _lib.ptr(out), _lib.ptr(h),
                                              _lib.current_stream()), "ope_agent_forward")
        return out, h[-1], no_sequence


class R_MADDPG_Actor(_RnnNet):
    def __init__(self, args, obs_dim, act_dim, device, cfg, take_prev_action=False, flat=None, values=None):
        if take_prev_action:
            raise NotImplementedError("prev_act_inp=True is not on the accelerated path")
        offs, sizes, total = rddpg_layout(cfg, 0)
        device = torch.device(device)
        if flat is None:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
        heads = None if np.ndim(act_dim) == 0 else [int(x) for x in act_dim]
        A = int(act_dim) if heads is None else int(sum(heads))
        if heads is None:
            names, shapes, o = _BODY + ["act.action_out.weight", "act.action_out.bias"], agent_param_shapes(int(obs_dim), A), offs
        else:      # multi-discrete: act.action_outs.{i}.{weight,bias} (act.py:14-17) = row ranges of the one stacked head the kernels see
            names, shapes, o, r0 = list(_BODY), agent_param_shapes(int(obs_dim), A)[:20], list(offs[:20]), 0
            for i, a_dim in enumerate(heads):
                names += ["act.action_outs.%d.weight" % i, "act.action_outs.%d.bias" % i]
                shapes += [(a_dim, H), (a_dim,)]
                o += [offs[20] + r0 * H, offs[21] + r0]
                r0 += a_dim
        super().__init__(names, shapes, o, flat)
        self.head_dims = heads
        self.obs_dim, self.act_dim, self.hidden_size, self.device, self.padded_numel = int(obs_dim), A, H, device, total
        self.take_prev_act = False
        self._setup_forward(obs_dim, A)
        if values is not None:
            for p, v in zip(list(self.parameters())[:20], values[:20]):
                p.data.copy_(v)
            flat[offs[20]:offs[20] + A * H].view(A, H).copy_(values[20])
            flat[offs[21]:offs[21] + A].copy_(values[21])

    def forward(self, obs, prev_acts, rnn_states):
        """Action logits and the new hidden state (r_actor_critic.py:36-67)."""
        out, h_final, no_sequence = self._run(obs, rnn_states)
        return (out[0] if no_sequence else out), h_final

    __call__ = forward


class R_MADDPG_Critic(_RnnNet):
    def __init__(self, args, central_obs_dim, central_act_dim, device, cfg, num_q_outs=1, flat=None, values=None):
        offs, sizes, total = rddpg_layout(cfg, 1)
        device = torch.device(device)
        if flat is None:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
        in_dim, K = int(central_obs_dim + central_act_dim), int(num_q_outs)
        names, shapes, o = list(_BODY), agent_param_shapes(in_dim, 1)[:20], offs[:20]
        for k in range(K):           # named_parameters() order: q_outs.0.weight, q_outs.0.bias, q_outs.1.weight, ...
            names += ["q_outs.%d.weight" % k, "q_outs.%d.bias" % k]
            shapes += [(1, H), (1,)]
            o = o + [offs[20] + k * H, offs[21] + k]
        super().__init__(names, shapes, o, flat)
        self.input_dim, self.num_q_outs, self.hidden_size, self.device, self.padded_numel = in_dim, K, H, device, total
        self._setup_forward(in_dim, K)
        if values is not None:
            for p, v in zip(list(self.parameters())[:20], values[:20]):
                p.data.copy_(v)
            flat[offs[20]:offs[20] + K * H].view(K, H).copy_(values[20])
            flat[offs[21]:offs[21] + K].copy_(values[21])

    def forward(self, central_obs, central_act, rnn_states):
        """List of Q-value tensors (one per head) and the new hidden state (r_actor_critic.py:93-129)."""
        co = torch.as_tensor(central_obs, dtype=torch.float32, device=self.device)
        ca = torch.as_tensor(central_act, dtype=torch.float32, device=self.device)
        out, h_final, no_sequence = self._run(torch.cat([co, ca], dim=-1), rnn_states)
        qs = [out[..., k:k + 1] for k in range(self.num_q_outs)]
        if no_sequence:
            qs = [q[0] for q in qs]
        return qs, h_final

    __call__ = forward
