"""Shared helpers for the -m gpu parity tests: build our buffer/policy/trainer from a golden fixture."""
import numpy as np
import torch

from conftest import sub
from golden_util import fixture_dims, fixture_episodes, EP_KEYS


def make_args(g, **over):
    from offpolicy_amd.config import default_args
    a = default_args(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]),
                     use_huber_loss=bool(g["hp_huber"]), huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]),
                     per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]), tau=float(g["hp_tau"]),
                     max_grad_norm=float(g["hp_maxnorm"]), use_double_q=bool(g["hp_double_q"]),
                     prev_act_inp=bool(g["hp_prev_act_inp"]) if "hp_prev_act_inp" in g else False,
                     use_same_share_obs=bool(g["hp_same_share"]) if "hp_same_share" in g else True)
    if "hp_gain" in g:
        a.gain, a.use_soft_update = float(g["hp_gain"]), bool(g["hp_soft_update"])
    if "hp_hypernet_layers" in g:      # network-shape flags (the engine supports hypernet_layers 1 | 2; the rest is refused loudly)
        a.hypernet_layers, a.layer_N, a.hidden_size = int(g["hp_hypernet_layers"]), int(g["hp_layer_N"]), int(g["hp_hidden_size"])
    if "hp_feature_norm" in g:
        a.use_feature_normalization = bool(g["hp_feature_norm"])
    if "hp_use_relu" in g:
        a.use_ReLU = bool(g["hp_use_relu"])
    for k, v in over.items():
        setattr(a, k, v)
    return a


def build_from_fixture(g, device="cuda:0"):
    """(dims, buffer, policy, trainer) with the fixture's initial weights loaded and its episodes inserted."""
    from offpolicy_amd.utils.synth import policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    dims = fixture_dims(g)
    args = make_args(g)
    md = [int(x) for x in g["multi_discrete"]] if "multi_discrete" in g else None      # sub-action sizes: no availability masks upstream
    pinfo = policy_info_for(dims, multi_discrete=md)
    n_pre = int(g["pre_idx_range"].shape[0]) if "pre_idx_range" in g else 0
    cap = max(int(g["filled_i"]), int(g["idx_range"].max()) + 1)
    buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, bool(args.use_same_share_obs), md is None,
                          False, device=device)
    if n_pre:
        d0 = as_policy_dicts({k: g["pre_ep/" + k] for k in EP_KEYS})
        r0 = buf.insert(n_pre, d0["obs"], d0["share_obs"], d0["acts"], d0["rewards"], d0["dones"], d0["dones_env"], d0["avail_acts"])
        assert np.array_equal(r0, g["pre_idx_range"])
    ep = fixture_episodes(g)
    d = as_policy_dicts(ep)
    r = buf.insert(len(g["idx_range"]), d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    assert np.array_equal(r, g["idx_range"])
    assert buf.policy_buffers["policy_0"].filled_i == int(g["filled_i"])
    assert buf.policy_buffers["policy_0"].current_i == int(g["current_i"])
    policy = QMixPolicy({"args": args, "device": torch.device(device)}, pinfo["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=torch.device(device),
                   episode_length=dims.episode_length, vdn=bool(g["vdn"]))
    policy.q_network.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()})
    if not bool(g["vdn"]):
        trainer.mixer.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, "mixer/").items()})
    trainer.hard_target_updates()
    return dims, buf, policy, trainer


def batch_from(buf, inds, weights=None):
    s = buf.policy_buffers["policy_0"].sample_inds(inds)
    return tuple({"policy_0": a} for a in s) + (weights, np.asarray(inds) if weights is not None else None)
