"""QMIX mixing network: parameter container (the arithmetic lives in csrc/ope_mixer.hip).

Mirror of offpolicy/algorithms/qmix/algorithm/q_mixer.py:6-94 (`QMixer`) for hypernet_layers = 2 (the default) and 1
(hyper_w1 / hyper_w2 as single Linear layers from the state, q_mixer.py:39-44): same parameter names / shapes / order and the
same constructor-time RNG consumption (nn.Linear default init followed by orthogonal_ / xavier_uniform_, biases 0:
q_mixer.py:33-66).
"""
import torch
import torch.nn as nn

from ....flat_module import FlatModule

MIX, HYP = 32, 64

MIXER_PARAM_NAMES = [
    "hyper_w1.0.weight", "hyper_w1.0.bias", "hyper_w1.2.weight", "hyper_w1.2.bias",
    "hyper_w2.0.weight", "hyper_w2.0.bias", "hyper_w2.2.weight", "hyper_w2.2.bias",
    "hyper_b1.weight", "hyper_b1.bias",
    "hyper_b2.0.weight", "hyper_b2.0.bias", "hyper_b2.2.weight", "hyper_b2.2.bias",
]


MIXER_PARAM_NAMES_1 = [        # hypernet_layers = 1
    "hyper_w1.weight", "hyper_w1.bias", "hyper_w2.weight", "hyper_w2.bias", "hyper_b1.weight", "hyper_b1.bias",
    "hyper_b2.0.weight", "hyper_b2.0.bias", "hyper_b2.2.weight", "hyper_b2.2.bias",
]


def mixer_param_shapes(n_agents, cent_obs_dim, hypernet_layers=2):
    N, S = n_agents, cent_obs_dim
    if hypernet_layers == 1:
        return [(N * MIX, S), (N * MIX,), (MIX, S), (MIX,), (MIX, S), (MIX,), (HYP, S), (HYP,), (1, HYP), (1,)]
    return [(HYP, S), (HYP,), (N * MIX, HYP), (N * MIX,), (HYP, S), (HYP,), (MIX, HYP), (MIX,),
            (MIX, S), (MIX,), (HYP, S), (HYP,), (1, HYP), (1,)]


def init_mixer_values(n_agents, cent_obs_dim, use_orthogonal=True, hypernet_layers=2):
    init_w = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
    N, S = n_agents, cent_obs_dim

    def lin(i, o):
        m = nn.Linear(i, o)
        init_w(m.weight.data)
        nn.init.constant_(m.bias.data, 0)
        return m
    if hypernet_layers == 1:      # construction order of q_mixer.py:39-66: hyper_w1, hyper_w2, hyper_b1, hyper_b2.0, hyper_b2.2
        mods = (lin(S, N * MIX), lin(S, MIX), lin(S, MIX), lin(S, HYP), lin(HYP, 1))
        out = []
        for m in mods:
            out += [m.weight.data.detach().float(), m.bias.data.detach().float()]
        return out
    w1a, w1b = lin(S, HYP), lin(HYP, N * MIX)
    w2a, w2b = lin(S, HYP), lin(HYP, MIX)
    b1 = lin(S, MIX)
    b2a, b2b = lin(S, HYP), lin(HYP, 1)
    out = []
    for m in (w1a, w1b, w2a, w2b, b1, b2a, b2b):
        out += [m.weight.data.detach().float(), m.bias.data.detach().float()]
    return out


class QMixer(FlatModule):
    def __init__(self, args, num_agents, cent_obs_dim, device, flat, offsets, multidiscrete_list=None, init=True):
        if multidiscrete_list:
            raise NotImplementedError("multi-discrete action spaces are not on the accelerated path")
        self.hypernet_layers = int(getattr(args, "hypernet_layers", 2))
        if self.hypernet_layers not in (1, 2):
            raise NotImplementedError("hypernet_layers must be 1 or 2 (q_mixer.py:39-57)")
        super().__init__(MIXER_PARAM_NAMES_1 if self.hypernet_layers == 1 else MIXER_PARAM_NAMES,
                         mixer_param_shapes(num_agents, cent_obs_dim, self.hypernet_layers), offsets, flat)
        self.device = torch.device(device)
        self.num_agents, self.cent_obs_dim = num_agents, cent_obs_dim
        self.hidden_layer_dim, self.hypernet_hidden_dim = MIX, HYP
        self.num_mixer_q_inps = num_agents
        if init:
            for p, v in zip(self.parameters(), init_mixer_values(num_agents, cent_obs_dim, getattr(args, "use_orthogonal", True), self.hypernet_layers)):
                p.data.copy_(v)


class VDNMixer(FlatModule):
    """VDN has no parameters (vdn_mixer.py:6-40): an empty module so `state_dict()` / `parameters()` still work."""

    def __init__(self, args, num_agents, cent_obs_dim, device, multidiscrete_list=None):
        super().__init__([], [], [], torch.empty(0, device=device))
        self.device = torch.device(device)
        self.num_agents = num_agents
        self.num_mixer_q_inps = num_agents
