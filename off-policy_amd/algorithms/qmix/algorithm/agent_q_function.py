"""Per-agent recurrent Q network: parameter container + forward through the HIP kernels.

Mirror of offpolicy/algorithms/qmix/algorithm/agent_q_function.py:8-67 (AgentQFunction) together with the bodies
it composes (algorithms/utils/{mlp,rnn,act}.py). Same `named_parameters()`, same
initialisation procedure and RNG consumption order as the reference's constructors, so `torch.manual_seed(s)`
followed by construction yields the reference's initial weights. The arithmetic is `ope_agent_forward`.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from .... import _lib
from ....flat_module import FlatModule

H = 64

AGENT_PARAM_NAMES = [
    "rnn.feature_norm.weight", "rnn.feature_norm.bias",
    "rnn.mlp.fc1.0.weight", "rnn.mlp.fc1.0.bias", "rnn.mlp.fc1.2.weight", "rnn.mlp.fc1.2.bias",
    "rnn.mlp.fc_h.0.weight", "rnn.mlp.fc_h.0.bias", "rnn.mlp.fc_h.2.weight", "rnn.mlp.fc_h.2.bias",
    "rnn.mlp.fc2.0.0.weight", "rnn.mlp.fc2.0.0.bias", "rnn.mlp.fc2.0.2.weight", "rnn.mlp.fc2.0.2.bias",
    "rnn.rnn.rnn.weight_ih_l0", "rnn.rnn.rnn.weight_hh_l0", "rnn.rnn.rnn.bias_ih_l0", "rnn.rnn.rnn.bias_hh_l0",
    "rnn.rnn.norm.weight", "rnn.rnn.norm.bias",
    "q.action_out.weight", "q.action_out.bias",
]


def agent_param_names(layer_N=1):
    """The tensors of the flat agent block, in layout order: named_parameters() of AgentQFunction with `layer_N` hidden blocks behind
    fc1 (mlp.py:14-28: fc2 = layer_N clones of fc_h). With use_feature_normalization = False the first two slots hold constants and are
    not parameters (exposed_agent_names)."""
    if layer_N == 1:
        return list(AGENT_PARAM_NAMES)
    i = AGENT_PARAM_NAMES.index("rnn.rnn.rnn.weight_ih_l0")
    extra = ["rnn.mlp.fc2.%d.%s" % (b, t) for b in range(1, layer_N) for t in ("0.weight", "0.bias", "2.weight", "2.bias")]
    return AGENT_PARAM_NAMES[:i] + extra + AGENT_PARAM_NAMES[i:]


def exposed_agent_names(layer_N=1, feature_norm=True):
    """named_parameters() as the reference's module has them: without rnn.feature_norm.* when MLPBase has no input LayerNorm
    (mlp.py:60-62)."""
    return [k for k in agent_param_names(layer_N) if feature_norm or not k.startswith("rnn.feature_norm.")]


def agent_param_shapes(obs_dim, act_dim, layer_N=1):
    D, A = obs_dim, act_dim
    return ([(D,), (D,), (H, D), (H,), (H,), (H,), (H, H), (H,), (H,), (H,)] + [(H, H), (H,), (H,), (H,)] * layer_N +
            [(3 * H, H), (3 * H, H), (3 * H,), (3 * H,), (H,), (H,), (A, H), (A,)])


def agent_layout(obs_dim, act_dim, layer_N=1):
    """(offsets, sizes, padded_total) of the agent block, from the library (single source of truth). The layout does not depend on
    use_feature_normalization: the two feature_norm slots stay (OPE_DIMS_NO_FEATURE_NORM, ope.h)."""
    cfg = _lib.QmixCfg()
    cfg.dims = _lib.Dims(1, act_dim, obs_dim, 1, 1, layer_N)
    cfg.batch = 1
    cfg.vdn = 1
    off = (C.c_int64 * 48)()
    siz = (C.c_int64 * 48)()
    total = _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz)
    if total < 0:
        _lib.check(int(total), "ope_qmix_param_layout")
    n = 22 + 4 * (layer_N - 1)
    return list(off)[:n], list(siz)[:n], int(total)


def init_agent_values(obs_dim, act_dim, use_orthogonal=True, gain_out=0.01, use_ReLU=True, layer_N=1, head_dims=None):
    """Initial values in named_parameters() order, drawn exactly as the reference's constructors draw them:
    nn.Linear default init then orthogonal_/xavier_uniform_ re-init (mlp.py:12-23, util.py:113-116), nn.GRU default
    init then per-parameter re-init (rnn.py:8-16), head with gain=args.gain (act.py:10-12). Biases 0, LayerNorms 1/0."""
    init_w = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
    gain = nn.init.calculate_gain("relu" if use_ReLU else "tanh")
    fc1 = nn.Linear(obs_dim, H)
    init_w(fc1.weight.data, gain=gain)
    fch = nn.Linear(H, H)
    init_w(fch.weight.data, gain=gain)
    gru = nn.GRU(H, H, num_layers=1)
    for name, p in gru.named_parameters():
        if "bias" in name:
            nn.init.constant_(p, 0)
        elif "weight" in name:
            init_w(p)
    if head_dims is None:
        qo = nn.Linear(H, act_dim)
        init_w(qo.weight.data, gain=gain_out)
        qw = qo.weight.data
    else:      # MultiDiscrete (act.py:14-17): one Linear per sub-action, constructed (and re-initialised) in order; stacked = the kernels' one head
        heads = []
        for d in head_dims:
            lin = nn.Linear(H, int(d))
            init_w(lin.weight.data, gain=gain_out)
            heads.append(lin.weight.data)
        qw = torch.cat(heads, dim=0)
        assert qw.shape[0] == act_dim
    one, zero = torch.ones, torch.zeros
    vals = [one(obs_dim), zero(obs_dim), fc1.weight.data, zero(H), one(H), zero(H),
            fch.weight.data, zero(H), one(H), zero(H),                       # fc_h (registered, unused)
            ] + [fch.weight.data.clone(), zero(H), one(H), zero(H)] * layer_N + [        # fc2[i] = deepcopy(fc_h) (mlp.py:23): identical clones
            gru.weight_ih_l0.data, gru.weight_hh_l0.data, zero(3 * H), zero(3 * H), one(H), zero(H),
            qw, zero(act_dim)]
    return [v.detach().float() for v in vals]


class AgentQFunction(FlatModule):
    def __init__(self, args, input_dim, act_dim, device, flat=None, _init=True):
        """`act_dim`: an int (Discrete), or the array of a MultiDiscrete space's sub-action sizes (agent_q_function.py:27 hands it to ACTLayer,
        act.py:14-17: one Linear head per sub-action). The kernels carry ONE stacked head of sum(act_dim) rows; this module exposes its row
        blocks under upstream's names `q.action_outs.{i}.weight / .bias` (same names, shapes and order in named_parameters() / state_dict())."""
        head_dims = None
        if np.ndim(act_dim) != 0:
            head_dims = [int(d) for d in np.asarray(act_dim).reshape(-1)]
            act_dim = int(sum(head_dims))
        self.head_dims = head_dims
        input_dim, act_dim, device = int(input_dim), int(act_dim), torch.device(device)
        self.layer_N = int(getattr(args, "layer_N", 1))
        if self.layer_N not in (1, 2):
            raise NotImplementedError("ope kernels support layer_N = 1 or 2 (got %r)" % self.layer_N)
        self.feature_norm = bool(getattr(args, "use_feature_normalization", True))
        self.use_relu = bool(getattr(args, "use_ReLU", True))
        if not self.use_relu and (self.layer_N != 1 or input_dim > 384):
            raise NotImplementedError("use_ReLU=False (tanh) on the accelerated path: one hidden block, network input width <= 384")
        offs, sizes, total = agent_layout(input_dim, act_dim, self.layer_N)
        own = flat is None
        if own:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
        names, shapes = agent_param_names(self.layer_N), agent_param_shapes(input_dim, act_dim, self.layer_N)
        keep = [i for i, k in enumerate(names) if self.feature_norm or not k.startswith("rnn.feature_norm.")]
        names, shapes, offs_k = [names[i] for i in keep], [shapes[i] for i in keep], [offs[i] for i in keep]
        if head_dims is not None:      # the stacked head's row blocks under upstream's per-head names (weight, bias per head, in order)
            assert names[-2:] == ["q.action_out.weight", "q.action_out.bias"]
            ow, ob = offs_k[-2], offs_k[-1]
            names, shapes, offs_k = names[:-2], shapes[:-2], offs_k[:-2]
            lo = 0
            for i, d in enumerate(head_dims):
                names += ["q.action_outs.%d.weight" % i, "q.action_outs.%d.bias" % i]
                shapes += [(d, H), (d,)]
                offs_k += [ow + lo * H, ob + lo]
                lo += d
        super().__init__(names, shapes, offs_k, flat)
        if not self.feature_norm:      # the two slots the kernels still read: gamma = 1, beta = 0, never trained (ope.h, OPE_DIMS_NO_FEATURE_NORM)
            with torch.no_grad():
                flat[offs[0]:offs[0] + input_dim] = 1.0
                flat[offs[1]:offs[1] + input_dim] = 0.0
        self.input_dim, self.act_dim, self.hidden_size, self.device = input_dim, act_dim, H, device
        self.padded_numel = total
        self._args = args
        if own and _init:
            vals = init_agent_values(input_dim, act_dim, getattr(args, "use_orthogonal", True),
                                     getattr(args, "gain", 0.01), getattr(args, "use_ReLU", True), self.layer_N, head_dims=head_dims)
            vals = [vals[i] for i in keep]
            if head_dims is not None:      # split the stacked head's values the way its rows are exposed
                qw, qb = vals[-2], vals[-1]
                vals = vals[:-2]
                lo = 0
                for d in head_dims:
                    vals += [qw[lo:lo + d], qb[lo:lo + d]]
                    lo += d
            for p, v in zip(self.parameters(), vals):
                p.data.copy_(v)
        self._dims = _lib.Dims(1, act_dim, input_dim, 1, 1, self.layer_N,
                               (0 if self.feature_norm else _lib.OPE_DIMS_NO_FEATURE_NORM) | (0 if self.use_relu else _lib.OPE_DIMS_TANH))
        self._ws = None

    def twin(self, flat):
        """Same structure bound to another flat vector (used for target networks)."""
        return AgentQFunction(self._args, self.input_dim, np.asarray(self.head_dims) if self.head_dims is not None else self.act_dim, self.device,
                              flat=flat, _init=False)

    def forward(self, obs, rnn_states):
        """q values for every action and the new hidden state (agent_q_function.py:34-67).
        obs [L, R, D] or [R, D]; rnn_states [R, H] (or [1, R, H]). Returns (q [L,R,A] or [R,A], h_final [R,H])."""
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self.device)
        rnn_states = torch.as_tensor(rnn_states, dtype=torch.float32, device=self.device)
        no_sequence = obs.dim() == 2
        if no_sequence:
            obs = obs[None]
        if rnn_states.dim() == 3:
            rnn_states = rnn_states[0]
        L, R = int(obs.shape[0]), int(obs.shape[1])
        obs = obs.contiguous()
        h0 = rnn_states.contiguous()
        need = _lib.lib.ope_agent_forward_workspace_bytes(C.byref(self._dims), L, R)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        q = torch.empty((L, R, self.act_dim), dtype=torch.float32, device=self.device)
        h = torch.empty((L, R, H), dtype=torch.float32, device=self.device)
        flat = self._flat
        _lib.check(_lib.lib.ope_agent_forward(C.byref(self._dims), L, R, _lib.ptr(obs), _lib.ptr(h0), _lib.ptr(flat),
                                              _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(q), _lib.ptr(h),
                                              _lib.current_stream()), "ope_agent_forward")
        h_final = h[-1]
        return (q[0] if no_sequence else q), h_final

    __call__ = forward
