"""Pin oracle/maddpg_oracle.py against outputs of the REAL reference (tests/golden/maddpg_*.npz, matd3_*.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from oracle import maddpg_oracle as DO
from oracle.qmix_oracle import HP
from test_mlp_oracle_golden import T_KEYS

CASES = ["maddpg_spread", "matd3_spread", "maddpg_small_huber_per", "matd3_small", "maddpg_small_wd",
         # round 4: continuous (Box) action spaces -- the action is the actor's output, MATD3's target noise is additive gaussian
         "maddpg_cont_small", "matd3_cont_small", "maddpg_cont_spread",
         # round 4: multi-discrete action spaces -- one Linear head, one argmax / gumbel-softmax per sub-action
         "maddpg_md_small", "matd3_md_small",
         # round 5: BASELINE config 3 at its own batch size (B = 256), stepped by the reference (oracle/make_golden_fullsize.py)
         "maddpg_spread_b256", "matd3_spread_b256"]


def ddpg_oracle_from(g):
    hp = HP(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
            huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_eps=float(g["hp_per_eps"]), tau=float(g["hp_tau"]),
            max_grad_norm=float(g["hp_maxnorm"]), weight_decay=float(g["hp_wd"]) if "hp_wd" in g else 0.0)
    return DO.MaddpgOracle(sub(g, "actor/"), sub(g, "critic/"), (g["heads/w"], g["heads/b"]), sub(g, "actor_tgt/"),
                           sub(g, "critic_tgt/"), (g["heads_tgt/w"], g["heads_tgt/b"]), int(g["dims"][0]), hp, td3=bool(g["td3"]),
                           continuous=bool(g["continuous"]) if "continuous" in g else False,
                           head_dims=[int(x) for x in g["multi_discrete"]] if "multi_discrete" in g else None)


def noise_for(g, step):
    """The uniform draws the reference consumed at `step` (torch.manual_seed(1000 + step), target noise first)."""
    n, a = int(g["dims"][0]), int(g["dims"][1])
    B = len(g["inds"])
    torch.manual_seed(1000 + step)
    if "continuous" in g and bool(g["continuous"]):      # gaussian_noise(shape, target_action_noise_std) for MATD3's target action; nothing else is drawn
        return (torch.empty(n * B, a).normal_(mean=0, std=0.2) if bool(g["td3"]) else None), None
    if "multi_discrete" in g:      # one uniform block per sub-action head, in order (MADDPGPolicy.py:75-77), side by side
        heads = [int(x) for x in g["multi_discrete"]]
        blocks = lambda: torch.cat([torch.FloatTensor(n * B, k).uniform_() for k in heads], dim=-1)
        u_t = blocks() if bool(g["td3"]) else None
        return u_t, blocks()
    u_t = torch.FloatTensor(n * B, a).uniform_() if bool(g["td3"]) else None
    u_a = torch.FloatTensor(n * B, a).uniform_()
    return u_t, u_a


CENT_CASES = ["maddpg_cent_small", "maddpg_cent_huber_per"]   # cent_train_policy_on_batch (per-agent centralized observations), oracle/make_golden_cent.py


@pytest.mark.parametrize("name", CASES + CENT_CASES)
def test_train_steps_match_reference(name):
    g = load_golden(name)
    orc = ddpg_oracle_from(g)
    batch = tuple(g["batch/" + k] for k in T_KEYS)
    w = g["per_weights"] if "per_weights" in g else None
    for s in range(len(g["critic_loss"])):
        u_t, u_a = noise_for(g, s)
        out = orc.train_step(batch, u_t, u_a, weights=w, per_agent_cent=name in CENT_CASES)
        np.testing.assert_allclose(out["critic_loss"], g["critic_loss"][s], rtol=3e-5)
        np.testing.assert_allclose(out["critic_grad_norm"], g["critic_grad_norm"][s], rtol=3e-5)
        np.testing.assert_allclose(out["actor_loss"], g["actor_loss"][s], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out["actor_grad_norm"], g["actor_grad_norm"][s], rtol=1e-4)
        if w is not None:
            np.testing.assert_allclose(out["priorities"], g["priorities"][s], rtol=3e-5)
    for grp, dst in (("final_actor/", orc.actor), ("final_critic/", orc.critic), ("final_actor_tgt/", orc.actor_tgt),
                     ("final_critic_tgt/", orc.critic_tgt)):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(dst[k].numpy(), ref, rtol=0, atol=2e-5, err_msg=grp + k)


@pytest.mark.parametrize("name", ["maddpg_spread", "matd3_small"])
def test_policy_construction_reproduces_reference_rng_stream(name):
    """actor, critic (+heads), target actor, target critic (+its OWN heads) drawn in the reference's order."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.algorithms.maddpg.algorithm.actor_critic import draw_actor_values, draw_critic_values
    g = load_golden(name)
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    K = 2 if bool(g["td3"]) else 1
    args = default_args()
    torch.manual_seed(1)
    np.random.seed(1)
    av = draw_actor_values(args, d, a)
    cv = draw_critic_values(args, s + n * a, K)
    draw_actor_values(args, d, a)
    tv = draw_critic_values(args, s + n * a, K)
    for v, (k, ref) in zip(av, sub(g, "actor/").items()):
        assert np.array_equal(v.numpy(), ref), k
    for v, (k, ref) in zip(cv[:14], sub(g, "critic/").items()):
        assert np.array_equal(v.numpy(), ref), k
    assert np.array_equal(cv[14].numpy(), g["heads/w"]) and np.array_equal(tv[14].numpy(), g["heads_tgt/w"])
    assert not np.array_equal(g["heads/w"], g["heads_tgt/w"])          # A-4: live and target heads differ forever


def mlp_multi_kinds(g):
    """Action-space kind of every policy: None (Discrete), ("md", sub-action sizes) or "cont" (`*_multi_kinds`, `*_multi_md` fixtures)."""
    P = len(g["groups"])
    if "kind_cont" not in g:
        return [None] * P
    return ["cont" if int(g["kind_cont"][i]) else (("md", [int(x) for x in g["kind_heads/%d" % i]]) if "kind_heads/%d" % i in g else None)
            for i in range(P)]


@pytest.mark.parametrize("name", ["maddpg_multi", "matd3_multi_per", "maddpg_multi_sl", "matd3_multi_actdims", "matd3_multi_kinds", "maddpg_multi_md"])
def test_multi_policy_train_steps_match_reference(name):
    """share_policy = False (oracle/make_golden_ddpg.py, OPE_GOLDEN_ONLY=multi): every policy updated in turn on its own batch with the
    joint action assembled from all policies (maddpg.py:40-80), soft updates after all of them (runner/mlp/base_runner.py:196-217)."""
    g = load_golden(name)
    groups, A, td3 = [int(x) for x in g["groups"]], int(g["A"]), bool(g["td3"])
    P = len(groups)
    As = [int(x) for x in g["act_dims"]] if "act_dims" in g else [A] * P      # policies may differ in their number of actions (`*_sl`, `*_actdims`)
    kinds = mlp_multi_kinds(g)                                                # ... and in the kind of their action space (`*_kinds`, `*_md`)

    def draw(k, target):      # what policy k's get_actions draws: gaussian (continuous targets), one uniform block per head, or one block
        rows = groups[k] * B
        if kinds[k] == "cont":
            return torch.empty(rows, As[k]).normal_(mean=0, std=0.2) if target else None
        if kinds[k] is not None:
            return torch.cat([torch.FloatTensor(rows, a).uniform_() for a in kinds[k][1]], dim=-1)
        return torch.FloatTensor(rows, As[k]).uniform_()
    hp = lambda: HP(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
                    huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_eps=float(g["hp_per_eps"]), tau=float(g["hp_tau"]),
                    max_grad_norm=float(g["hp_maxnorm"]))
    pols = [DO.MaddpgOracle(sub(g, "p%d/actor/" % i), sub(g, "p%d/critic/" % i), (g["p%d/heads/w" % i], g["p%d/heads/b" % i]),
                            sub(g, "p%d/actor_tgt/" % i), sub(g, "p%d/critic_tgt/" % i), (g["p%d/heads_tgt/w" % i], g["p%d/heads_tgt/b" % i]),
                            groups[i], hp(), td3=td3, continuous=kinds[i] == "cont", head_dims=kinds[i][1] if isinstance(kinds[i], tuple) else None)
            for i in range(P)]
    multi = DO.MaddpgMultiOracle(pols)
    batches = [tuple(g["p%d/batch/%s" % (i, k)] for k in T_KEYS) for i in range(P)]
    B = len(g["inds"])
    w = g["per_weights"] if "per_weights" in g else None
    for s in range(len(g["p0/critic_loss"])):
        for i in range(P):
            torch.manual_seed(1000 + 10 * s + i)         # the reference's draws: target noise per policy in policy order, then the actor's
            u_ts = [draw(k, True) for k in range(P)] if td3 else None
            u_a = draw(i, False)
            out = multi.train_step(i, batches, u_ts, u_a, weights=w)
            np.testing.assert_allclose(out["critic_loss"], g["p%d/critic_loss" % i][s], rtol=3e-5, err_msg="%d p%d" % (s, i))
            np.testing.assert_allclose(out["critic_grad_norm"], g["p%d/critic_grad_norm" % i][s], rtol=3e-5)
            np.testing.assert_allclose(out["actor_loss"], g["p%d/actor_loss" % i][s], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(out["actor_grad_norm"], g["p%d/actor_grad_norm" % i][s], rtol=1e-4)
            if w is not None:
                np.testing.assert_allclose(out["priorities"], g["p%d/priorities" % i][s], rtol=3e-5)
        multi.soft_target_updates()
    for i, o in enumerate(pols):
        for grp, dst in (("actor/", o.actor), ("critic/", o.critic), ("actor_tgt/", o.actor_tgt), ("critic_tgt/", o.critic_tgt)):
            for k, ref in sub(g, "final/p%d/%s" % (i, grp)).items():
                np.testing.assert_allclose(dst[k].numpy(), ref, rtol=0, atol=2e-5, err_msg="p%d %s%s" % (i, grp, k))
