"""CPU: the rollout-side host logic of the MADDPG-family policies -- everything of get_actions behind the actor network, and
get_random_actions -- against what the REFERENCE returned (tests/golden/rollout_actions.npz, oracle/make_golden_rollout.py): Discrete,
MultiDiscrete and Box action spaces, MLP and recurrent, MADDPG and MATD3, exploring / greedy / target / gumbel calls, with and without
availability masks. The engine's `_actions_from_actor_out` runs on the reference's recorded actor output under the recorded numpy and torch
generator states: the same generators must be consumed in the same order, so the actions are compared exactly (the continuous ones to
float rounding). The actor network itself is covered by the GPU tests."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden


def _stub(g, family, kind):
    from offpolicy_amd.utils.spaces import Discrete, Box, MultiDiscrete
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import DecayThenFlatSchedule
    eps0, eps1, eps_t, act_noise, tgt_noise = [float(x) for x in g["hp"]]
    td3 = "_td3_" in family + "_"
    heads = [int(x) for x in g[family + "_" + kind + "/heads"]]
    width = int(g[family + "_" + kind + "/width"][0])
    space = {"discrete": Discrete(width), "md": MultiDiscrete([[0, k - 1] for k in heads]) if heads else None,
             "cont": Box(low=-np.ones(width, np.float32), high=np.ones(width, np.float32))}[kind]
    return SimpleNamespace(discrete=kind != "cont", multidiscrete=kind == "md", act_dim=np.array(heads) if kind == "md" else width,
                           output_dim=width, act_space=space, args=SimpleNamespace(act_noise_std=act_noise),
                           target_noise=tgt_noise if td3 else None,
                           exploration=DecayThenFlatSchedule(eps0, eps1, eps_t, decay="linear"))


def _set_rng(g, pre):
    np.random.set_state(("MT19937", g[pre + "np_keys"], int(g[pre + "np_pos"][0]), int(g[pre + "np_pos"][1]), float(g[pre + "np_gauss"][0])))
    torch.set_rng_state(torch.from_numpy(g[pre + "torch"]))


def test_rollout_post_processing_matches_the_reference_call_by_call():
    from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy
    from offpolicy_amd.algorithms.r_maddpg.algorithm.rMADDPGPolicy import R_MADDPGPolicy
    g = load_golden("rollout_actions")
    names = [str(x) for x in g["names"]]
    assert len(names) == 68
    seen = set()
    for name in names:
        pre = name + "/"
        head = name.split("/")[0]                       # e.g. rnn_td3_md
        fam, algo, kind = head.split("_")
        stub = _stub(g, fam + "_" + algo, kind)
        is_random, explore, use_target, use_gumbel, t_env = [int(x) for x in g[pre + "flags"]]
        avail = g[pre + "avail"] if pre + "avail" in g else None
        out = torch.as_tensor(g[pre + "actor_out"])
        B = out.shape[0]
        _set_rng(g, pre)
        if is_random:
            fn = MADDPGPolicy.get_random_actions if fam == "mlp" else R_MADDPGPolicy.get_random_actions
            got, eps = fn(stub, g[pre + "obs"], avail), None
        elif fam == "mlp":
            got, eps = MADDPGPolicy._actions_from_actor_out(stub, out, B, avail, t_env if t_env >= 0 else None, bool(explore), bool(use_target),
                                                            bool(use_gumbel))
        else:
            got, eps = R_MADDPGPolicy._actions_from_actor_out(stub, out, B, True, avail, t_env if t_env >= 0 else None, bool(explore),
                                                              bool(use_target), bool(use_gumbel))
        got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
        ref = g[pre + "actions"]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        if kind == "cont":
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6, err_msg=name)
        else:
            assert np.array_equal(got.astype(np.float64), ref.astype(np.float64)), name
        ref_eps = float(g[pre + "eps"][0])
        assert (eps is None) == (ref_eps < 0), name
        if eps is not None:
            np.testing.assert_allclose(float(eps), ref_eps, rtol=1e-12, err_msg=name)
        seen.add((fam, algo, kind, bool(is_random), bool(explore), bool(use_target), bool(use_gumbel)))
    assert len(seen) == 56      # every (family, algorithm, kind, call flavour) combination was exercised (the two exploring calls differ in t_env only)


def test_q_learning_rollout_post_processing_matches_the_reference_call_by_call():
    """The Q-learning policies (QMixPolicy, M_QMixPolicy; round 5: also MultiDiscrete action spaces, one greedy choice and one pair of
    exploration draws per sub-action head): actions_from_q / get_random_actions on the reference network's recorded q values under the
    recorded generator states (tests/golden/rollout_actions_q.npz, oracle/make_golden_rollout_q.py) -- the one-hot actions and the greedy
    q values the reference returned, exactly."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.spaces import Discrete, MultiDiscrete
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy
    g = load_golden("rollout_actions_q")
    eps0, eps1, eps_t = [float(x) for x in g["hp"]]
    args = default_args(epsilon_start=eps0, epsilon_finish=eps1, epsilon_anneal_time=eps_t)
    names = [str(x) for x in g["names"]]
    assert len(names) == 16
    pols = {}
    for name in names:
        pre = name + "/"
        fam, kind = name.split("/")[0].split("_")
        heads = [int(x) for x in g[fam + "_" + kind + "/heads"]]
        width = int(g[fam + "_" + kind + "/width"][0])
        if (fam, kind) not in pols:
            space = MultiDiscrete([[0, k - 1] for k in heads]) if kind == "md" else Discrete(width)
            pols[(fam, kind)] = (QMixPolicy if fam == "rnn" else M_QMixPolicy)(
                {"args": args, "device": "cpu"}, {"cent_obs_dim": 9, "cent_act_dim": 2 * width, "obs_space": [6], "share_obs_space": [9], "act_space": space})
        pol = pols[(fam, kind)]
        is_random, explore, t_env = [int(x) for x in g[pre + "flags"]]
        avail = g[pre + "avail"] if pre + "avail" in g else None
        q = torch.as_tensor(g[pre + "q"])
        B = q.shape[0]
        qq = list(torch.split(q, heads, dim=-1)) if kind == "md" else q
        _set_rng(g, pre)
        if is_random:
            got, gq = pol.get_random_actions(np.zeros((B, 6), np.float32), avail), None
        elif fam == "rnn":
            got, gq = pol.actions_from_q(qq, available_actions=avail, explore=bool(explore), t_env=t_env if t_env >= 0 else None)
        else:
            got, gq = pol.actions_from_q(qq, B, avail, t_env if t_env >= 0 else None, bool(explore))
        ref = g[pre + "actions"]
        assert np.asarray(got).shape == ref.shape and np.array_equal(np.asarray(got, dtype=np.float64), ref.astype(np.float64)), name
        if gq is not None:
            np.testing.assert_array_equal(gq.detach().numpy(), g[pre + "greedy_q"], err_msg=name)
