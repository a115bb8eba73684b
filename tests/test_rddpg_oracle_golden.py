"""Pin oracle/rmaddpg_oracle.py against outputs of the REAL reference (tests/golden/rmaddpg_*.npz, rmatd3_*.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from golden_util import EP_KEYS
from oracle import rmaddpg_oracle as RO
from oracle.qmix_oracle import HP

CASES = ["rmaddpg_tiny", "rmatd3_tiny", "rmaddpg_odd_huber_per", "rmatd3_odd_per", "rmaddpg_3m",
         # round 4: continuous (Box) action spaces (rMADDPGPolicy.py:121-129)
         "rmaddpg_cont_tiny", "rmatd3_cont_odd",
         # round 4: multi-discrete action spaces (rMADDPGPolicy.py:81-102)
         "rmaddpg_md_tiny", "rmatd3_md_odd"]


def rddpg_oracle_from(g):
    hp = HP(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
            huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]),
            tau=float(g["hp_tau"]), max_grad_norm=float(g["hp_maxnorm"]))
    return RO.RMaddpgOracle(sub(g, "actor/"), sub(g, "critic/"), sub(g, "actor_tgt/"), sub(g, "critic_tgt/"), int(g["dims"][0]), hp,
                            td3=bool(g["td3"]), continuous=bool(g["continuous"]) if "continuous" in g else False,
                            head_dims=[int(x) for x in g["multi_discrete"]] if "multi_discrete" in g else None)


def rnoise_for(g, step, update_actor=True):
    """The uniform draws the reference consumed at `step` (torch.manual_seed(1000 + step)): target noise [T+1, N*B, A]
    first (MATD3 only), then actor noise [T, N*B, A] when the actor is updated."""
    n, a, _, _, T = [int(x) for x in g["dims"]]
    B = len(g["inds"])
    torch.manual_seed(1000 + step)
    if "continuous" in g and bool(g["continuous"]):      # gaussian_noise(shape, target_action_noise_std) of the target action (R_MATD3); nothing for the actor
        return (torch.empty(T + 1, n * B, a).normal_(mean=0, std=0.2) if bool(g["td3"]) else None), None
    if "multi_discrete" in g:      # one uniform block per sub-action head, in order, side by side
        heads = [int(x) for x in g["multi_discrete"]]
        blocks = lambda L: torch.cat([torch.FloatTensor(L, n * B, k).uniform_() for k in heads], dim=-1)
        u_t = blocks(T + 1) if bool(g["td3"]) else None
        return u_t, (blocks(T) if update_actor else None)
    u_t = torch.FloatTensor(T + 1, n * B, a).uniform_() if bool(g["td3"]) else None
    u_a = torch.FloatTensor(T, n * B, a).uniform_() if update_actor else None
    return u_t, u_a


FULLSIZE_CASES = ["rmatd3_MMM2_b128_per"]      # round 5: BASELINE config 5 at its own size (oracle/make_golden_fullsize.py; inputs regenerated)


def fixture_batch_np(g):
    """What `sample_inds` returned to the reference: stored for the small cases, rebuilt from the regenerated episodes otherwise
    (rec_buffer.py:192-240: x[:, inds] then the [N, T(+1), B, .] transpose; share_obs without the agent axis) and checked by digest."""
    if "batch/obs" in g:
        return tuple(g["batch/" + k] for k in EP_KEYS)
    from golden_util import rddpg_fixture_episodes, batch_digest
    ep = rddpg_fixture_episodes(g)
    inds = np.asarray(g["inds"])
    out = []
    for k in EP_KEYS:
        x = ep[k][:, inds]
        if k == "share_obs":
            x = x[:, :, 0]
        if k in ("share_obs", "dones_env"):
            out.append(np.ascontiguousarray(x))
        else:
            out.append(np.ascontiguousarray(x.transpose(2, 0, 1, 3)))
    assert batch_digest(out) == str(g["batch_digest"]), "rebuilt batch differs from what the reference's sample_inds returned"
    return tuple(out)


CENT_CASES = ["rmaddpg_cent_tiny", "rmatd3_cent_odd"]      # cent_train_policy_on_batch (per-agent centralized observations), oracle/make_golden_cent.py


@pytest.mark.parametrize("name", CASES + CENT_CASES + FULLSIZE_CASES)
def test_train_steps_match_reference(name):
    g = load_golden(name)
    orc = rddpg_oracle_from(g)
    batch = fixture_batch_np(g)
    w = g["per_weights"] if "per_weights" in g else None
    for s in range(len(g["critic_loss"])):
        upd = bool(g["update_actor"][s])
        u_t, u_a = rnoise_for(g, s, upd)
        out = orc.train_step(batch, u_t, u_a, weights=w, per_agent_cent=name in CENT_CASES)
        assert out["update_actor"] == upd
        np.testing.assert_allclose(out["critic_loss"], g["critic_loss"][s], rtol=3e-5)
        np.testing.assert_allclose(out["critic_grad_norm"], g["critic_grad_norm"][s], rtol=5e-5)
        if upd:
            np.testing.assert_allclose(out["actor_loss"], g["actor_loss"][s], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(out["actor_grad_norm"], g["actor_grad_norm"][s], rtol=2e-4)
        if w is not None:
            np.testing.assert_allclose(out["priorities"], g["priorities"][s], rtol=3e-5)
    for grp, dst in (("final_actor/", orc.actor), ("final_critic/", orc.critic), ("final_actor_tgt/", orc.actor_tgt),
                     ("final_critic_tgt/", orc.critic_tgt)):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(dst[k].numpy(), ref, rtol=0, atol=2e-5, err_msg=grp + k)


@pytest.mark.parametrize("name", ["rmaddpg_tiny", "rmatd3_tiny"])
def test_policy_construction_reproduces_reference_rng_stream(name):
    """actor, critic (K registered heads), target actor, target critic drawn in the reference's order."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.algorithms.r_maddpg.algorithm.r_actor_critic import draw_ractor_values, draw_rcritic_values
    g = load_golden(name)
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    K = 2 if bool(g["td3"]) else 1
    args = default_args()
    torch.manual_seed(1)
    np.random.seed(1)
    av = draw_ractor_values(args, d, a)
    cv = draw_rcritic_values(args, s + n * a, K)
    for v, (k, ref) in zip(av, sub(g, "actor/").items()):
        assert np.array_equal(v.numpy(), ref), k
    crit = sub(g, "critic/")
    for v, (k, ref) in zip(cv[:20], list(crit.items())[:20]):
        assert np.array_equal(v.numpy(), ref), k
    for k in range(K):
        assert np.array_equal(cv[20][k].numpy(), crit["q_outs.%d.weight" % k].reshape(-1))
    # the targets take the live weights (rMADDPGPolicy.py:50-51)
    for k, ref in sub(g, "critic_tgt/").items():
        assert np.array_equal(ref, crit[k])


MULTI_CASES = ["rmaddpg_multi_odd", "rmatd3_multi_tiny", "rmaddpg_multi_hetero", "rmatd3_multi_hetero", "rmaddpg_multi_sl", "rmatd3_multi_actdims",
               "rmatd3_multi_kinds", "rmaddpg_multi_md"]


def multi_policy_ids(g):
    return ["policy_%d" % i for i in range(len(g["groups"]))]


def multi_obs_dims(g):
    """Observation width of every policy (fixtures of round 3 on carry `obs_dims`; older ones share dims.obs_dim)."""
    return [int(x) for x in g["obs_dims"]] if "obs_dims" in g else [int(g["dims"][2])] * len(g["groups"])


def multi_act_dims(g):
    """Number of actions of every policy (`act_dims`: fixtures whose policies differ in it -- simple_speaker_listener's 3 and 5)."""
    return [int(x) for x in g["act_dims"]] if "act_dims" in g else [int(g["dims"][1])] * len(g["groups"])


def multi_kinds(g):
    """Action-space kind of every policy: None (Discrete), ("md", sub-action sizes) or "cont" (`*_multi_kinds`, `*_multi_md` fixtures)."""
    P = len(g["groups"])
    if "kind_cont" not in g:
        return [None] * P
    return ["cont" if int(g["kind_cont"][i]) else (("md", [int(x) for x in g["kind_heads/%d" % i]]) if "kind_heads/%d" % i in g else None)
            for i in range(P)]


def multi_batches(g):
    """Per-policy sample_inds 7-tuples rebuilt from the stored episodes: agent fields sliced to the policy's agents."""
    groups = [int(x) for x in g["groups"]]
    starts = np.cumsum([0] + groups)[:-1]
    inds = np.asarray(g["inds"])
    N = int(g["dims"][0])
    obs_dims, act_dims = multi_obs_dims(g), multi_act_dims(g)
    out = []
    for pi, (s0, n, od, ad) in enumerate(zip(starts, groups, obs_dims, act_dims)):
        fields = []
        for k in EP_KEYS:
            v = g["ep/" + k][:, inds]                                   # [T(+1), B, N, dim] or [T(+1), B, dim]
            if k == "acts" and "pol_acts/%d" % pi in g:                 # multi-discrete / continuous policies: their own stored actions
                fields.append(np.ascontiguousarray(g["pol_acts/%d" % pi][:, inds].transpose(2, 0, 1, 3)))
                continue
            if k == "obs":
                v = v[..., :od]                                         # policies may differ in observation width
            if k in ("acts", "avail_acts"):
                v = v[..., :ad]                                         # ... and in their number of actions
            if v.ndim == 4 and v.shape[2] == N and k != "share_obs":
                fields.append(np.ascontiguousarray(v[:, :, s0:s0 + n].transpose(2, 0, 1, 3)))
            elif k == "share_obs":
                fields.append(np.ascontiguousarray(v[:, :, 0] if v.ndim == 4 else v))
            else:
                fields.append(np.ascontiguousarray(v))
        out.append(tuple(fields))
    return out


def multi_noise(g, step, pi, update_actor):
    """Draws of one train call (torch.manual_seed(1000 + step*P + pi)): target noise of EVERY policy in policy order
    (get_update_info), then the update policy's actor noise."""
    _, a, _, _, T = [int(x) for x in g["dims"]]
    groups = [int(x) for x in g["groups"]]
    B = len(g["inds"])
    torch.manual_seed(1000 + step * len(groups) + pi)
    ads, kinds = multi_act_dims(g), multi_kinds(g)

    def draw(L, n, ad, kind, target):      # what that policy's get_actions draws: gaussian (continuous targets), one uniform block per head, or one block
        if kind == "cont":
            return torch.empty(L, n * B, ad).normal_(mean=0, std=0.2) if target else None
        if kind is not None:
            return torch.cat([torch.FloatTensor(L, n * B, k).uniform_() for k in kind[1]], dim=-1)
        return torch.FloatTensor(L, n * B, ad).uniform_()
    u_t = [draw(T + 1, n, ad, kd, True) for n, ad, kd in zip(groups, ads, kinds)] if bool(g["td3"]) else None
    u_a = draw(T, groups[pi], ads[pi], kinds[pi], False) if update_actor else None
    return u_t, u_a


def multi_oracle_from(g):
    hp = HP(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
            huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]),
            tau=float(g["hp_tau"]), max_grad_norm=float(g["hp_maxnorm"]))
    pol = [RO.RMaddpgOracle(sub(g, p + "/actor/"), sub(g, p + "/critic/"), sub(g, p + "/actor_tgt/"), sub(g, p + "/critic_tgt/"), int(n), hp,
                            td3=bool(g["td3"]), continuous=kd == "cont", head_dims=kd[1] if isinstance(kd, tuple) else None)
           for p, n, kd in zip(multi_policy_ids(g), g["groups"], multi_kinds(g))]
    return RO.RMaddpgMultiOracle(pol)


@pytest.mark.parametrize("name", MULTI_CASES)
def test_multi_policy_train_steps_match_reference(name):
    """share_policy = False (scripts/train_mpe_rmaddpg.sh): per step every policy is updated in turn; each update uses every
    policy's target actor for the joint target action and only replaces its own agents' blocks in the actor update."""
    g = load_golden(name)
    orc = multi_oracle_from(g)
    batches = multi_batches(g)
    pids = multi_policy_ids(g)
    for s in range(g["critic_loss"].shape[0]):
        for pi in range(len(pids)):
            upd = bool(g["update_actor"][s, pi])
            u_t, u_a = multi_noise(g, s, pi, upd)
            out = orc.train_step(pi, batches, u_t, u_a)
            assert out["update_actor"] == upd
            np.testing.assert_allclose(out["critic_loss"], g["critic_loss"][s, pi], rtol=3e-5)
            np.testing.assert_allclose(out["critic_grad_norm"], g["critic_grad_norm"][s, pi], rtol=5e-5)
            if upd:
                np.testing.assert_allclose(out["actor_loss"], g["actor_loss"][s, pi], rtol=1e-4, atol=1e-6)
                np.testing.assert_allclose(out["actor_grad_norm"], g["actor_grad_norm"][s, pi], rtol=2e-4)
    for p, o in zip(pids, orc.pol):
        for grp, dst in (("final_actor/", o.actor), ("final_critic/", o.critic), ("final_actor_tgt/", o.actor_tgt), ("final_critic_tgt/", o.critic_tgt)):
            for k, ref in sub(g, p + "/" + grp).items():
                np.testing.assert_allclose(dst[k].numpy(), ref, rtol=0, atol=2e-5, err_msg=p + "/" + grp + k)
