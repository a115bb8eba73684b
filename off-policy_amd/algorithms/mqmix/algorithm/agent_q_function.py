"""MLP per-agent Q network (mirror of offpolicy/algorithms/mqmix/algorithm/agent_q_function.py:8-41): MLPBase trunk +
Linear head, parameters under `mlp.*` / `q.*`. Same init RNG stream as the reference's constructor."""
import ctypes as C

import torch
import torch.nn as nn

from .... import _lib
from ....flat_module import FlatModule

H = 64

MLP_AGENT_PARAM_NAMES = [
    "mlp.feature_norm.weight", "mlp.feature_norm.bias",
    "mlp.mlp.fc1.0.weight", "mlp.mlp.fc1.0.bias", "mlp.mlp.fc1.2.weight", "mlp.mlp.fc1.2.bias",
    "mlp.mlp.fc_h.0.weight", "mlp.mlp.fc_h.0.bias", "mlp.mlp.fc_h.2.weight", "mlp.mlp.fc_h.2.bias",
    "mlp.mlp.fc2.0.0.weight", "mlp.mlp.fc2.0.0.bias", "mlp.mlp.fc2.0.2.weight", "mlp.mlp.fc2.0.2.bias",
    "q.action_out.weight", "q.action_out.bias",
]


def mlp_agent_param_shapes(obs_dim, act_dim):
    D, A = obs_dim, act_dim
    return [(D,), (D,), (H, D), (H,), (H,), (H,), (H, H), (H,), (H,), (H,), (H, H), (H,), (H,), (H,), (A, H), (A,)]


def mlp_agent_layout(obs_dim, act_dim):
    cfg = _lib.QmixCfg()
    cfg.dims = _lib.Dims(1, act_dim, obs_dim, 1, 1)
    cfg.batch, cfg.vdn, cfg.mlp = 1, 1, 1
    off, siz = (C.c_int64 * 48)(), (C.c_int64 * 48)()
    total = _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz)
    if total < 0:
        _lib.check(int(total), "ope_qmix_param_layout")
    return list(off)[:16], list(siz)[:16], int(total)


def init_mlp_agent_values(obs_dim, act_dim, use_orthogonal=True, gain_out=0.01, use_ReLU=True, head_dims=None):
    """MLPBase (mlp.py:52-74) then ACTLayer (act.py:5-19), drawn in the reference's order."""
    init_w = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
    gain = nn.init.calculate_gain("relu" if use_ReLU else "tanh")
    fc1 = nn.Linear(obs_dim, H)
    init_w(fc1.weight.data, gain=gain)
    fch = nn.Linear(H, H)
    init_w(fch.weight.data, gain=gain)
    if head_dims is None:
        qo = nn.Linear(H, act_dim)
        init_w(qo.weight.data, gain=gain_out)
        qw = qo.weight.data
    else:      # MultiDiscrete (act.py:14-17): one Linear per sub-action, in order; stacked = the kernels' one head
        heads = []
        for d in head_dims:
            lin = nn.Linear(H, int(d))
            init_w(lin.weight.data, gain=gain_out)
            heads.append(lin.weight.data)
        qw = torch.cat(heads, dim=0)
    one, zero = torch.ones, torch.zeros
    vals = [one(obs_dim), zero(obs_dim), fc1.weight.data, zero(H), one(H), zero(H), fch.weight.data, zero(H), one(H), zero(H),
            fch.weight.data.clone(), zero(H), one(H), zero(H), qw, zero(act_dim)]
    return [v.detach().float() for v in vals]


class AgentQFunction(FlatModule):
    def __init__(self, args, input_dim, act_dim, device, flat=None, _init=True):
        """`act_dim`: an int, or the array of a MultiDiscrete space's sub-action sizes: the one stacked head's row blocks are then exposed
        under upstream's names `q.action_outs.{i}.weight / .bias` (act.py:14-17), as in the recurrent module."""
        import numpy as np
        head_dims = None
        if np.ndim(act_dim) != 0:
            head_dims = [int(d) for d in np.asarray(act_dim).reshape(-1)]
            act_dim = int(sum(head_dims))
        self.head_dims = head_dims
        input_dim, act_dim, device = int(input_dim), int(act_dim), torch.device(device)
        # use_feature_normalization = False (mlp.py:60-62) / use_ReLU = False (mlp.py:9-12), as the recurrent module: the two feature_norm
        # slots of the flat vector stay (ones / zeros, not parameters: OPE_DIMS_NO_FEATURE_NORM), tanh rides on OPE_DIMS_TANH (input <= 384)
        self.feature_norm = bool(getattr(args, "use_feature_normalization", True))
        self.use_relu = bool(getattr(args, "use_ReLU", True))
        if not self.use_relu and input_dim > 384:
            raise NotImplementedError("use_ReLU=False (tanh) on the accelerated path: network input width <= 384")
        offs, sizes, total = mlp_agent_layout(input_dim, act_dim)
        offs_all = list(offs)
        own = flat is None
        if own:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
        names, shapes = list(MLP_AGENT_PARAM_NAMES), mlp_agent_param_shapes(input_dim, act_dim)
        keep = [i for i, k in enumerate(names) if self.feature_norm or not k.startswith("mlp.feature_norm.")]
        names, shapes, offs = [names[i] for i in keep], [shapes[i] for i in keep], [offs[i] for i in keep]
        if head_dims is not None:
            ow, ob = offs[-2], offs[-1]
            names, shapes, offs = names[:-2], shapes[:-2], list(offs[:-2])
            lo = 0
            for i, d in enumerate(head_dims):
                names += ["q.action_outs.%d.weight" % i, "q.action_outs.%d.bias" % i]
                shapes += [(d, H), (d,)]
                offs += [ow + lo * H, ob + lo]
                lo += d
        super().__init__(names, shapes, offs, flat)
        if not self.feature_norm:
            with torch.no_grad():
                flat[offs_all[0]:offs_all[0] + input_dim] = 1.0
                flat[offs_all[1]:offs_all[1] + input_dim] = 0.0
        self.input_dim, self.act_dim, self.hidden_size, self.device = input_dim, act_dim, H, device
        self.padded_numel = total
        self._args = args
        if own and _init:
            vals = init_mlp_agent_values(input_dim, act_dim, getattr(args, "use_orthogonal", True), getattr(args, "gain", 0.01),
                                         getattr(args, "use_ReLU", True), head_dims=head_dims)
            vals = [vals[i] for i in keep]
            if head_dims is not None:
                qw, qb = vals[-2], vals[-1]
                vals = vals[:-2]
                lo = 0
                for d in head_dims:
                    vals += [qw[lo:lo + d], qb[lo:lo + d]]
                    lo += d
            for p, v in zip(self.parameters(), vals):
                p.data.copy_(v)
        self._dims = _lib.Dims(1, act_dim, input_dim, 1, 1, 1,
                               (0 if self.feature_norm else _lib.OPE_DIMS_NO_FEATURE_NORM) | (0 if self.use_relu else _lib.OPE_DIMS_TANH))
        self._ws = None

    def twin(self, flat):
        import numpy as np
        return AgentQFunction(self._args, self.input_dim, np.asarray(self.head_dims) if self.head_dims is not None else self.act_dim, self.device,
                              flat=flat, _init=False)

    def forward(self, x):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        rows = int(x.shape[0])
        need = _lib.lib.ope_agent_forward_mlp_workspace_bytes(C.byref(self._dims), rows)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        q = torch.empty((rows, self.act_dim), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib.ope_agent_forward_mlp(C.byref(self._dims), rows, _lib.ptr(x), _lib.ptr(self._flat), _lib.ptr(self._ws),
                                                  self._ws.numel(), _lib.ptr(q), _lib.current_stream()), "ope_agent_forward_mlp")
        return q

    __call__ = forward
