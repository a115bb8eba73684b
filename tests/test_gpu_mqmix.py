"""-m gpu: MLP QMIX / VDN (M_QMix, M_VDN) and MlpReplayBuffer through the C-ABI vs the reference's frozen outputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from test_mlp_oracle_golden import T_KEYS, CASES

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def build(g, device="cuda:0"):
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims, policy_info_for
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy
    from offpolicy_amd.algorithms.mqmix.mqmix import M_QMix
    from offpolicy_amd.algorithms.mvdn.mvdn import M_VDN
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    dims = EnvDims("fx", n, a, d, s, 1)
    args = default_args(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]),
                        use_huber_loss=bool(g["hp_huber"]), huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]),
                        per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]), tau=float(g["hp_tau"]),
                        max_grad_norm=float(g["hp_maxnorm"]), use_double_q=bool(g["hp_double_q"]))
    args.use_ReLU = bool(g["hp_use_relu"]) if "hp_use_relu" in g else True
    args.use_feature_normalization = "agent/mlp.feature_norm.weight" in g
    pinfo = policy_info_for(dims)
    buf = MlpReplayBuffer(pinfo, {"policy_0": list(range(n))}, int(g["cap"]), True, True, False, device=device)
    wrap = lambda pre: [{"policy_0": g[pre + k]} for k in T_KEYS]
    if "pre_idx_range" in g:
        assert np.array_equal(buf.insert(len(g["pre_idx_range"]), *wrap("pre_tr/")), g["pre_idx_range"])
    assert np.array_equal(buf.insert(len(g["idx_range"]), *wrap("tr/")), g["idx_range"])
    pb = buf.policy_buffers["policy_0"]
    assert pb.filled_i == int(g["filled_i"]) and pb.current_i == int(g["current_i"])
    dev = torch.device(device)
    policy = M_QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
    if bool(g["vdn"]):
        trainer = M_VDN(args, n, {"policy_0": policy}, lambda x: "policy_0", device=dev)
    else:
        trainer = M_QMix(args, n, {"policy_0": policy}, lambda x: "policy_0", device=dev)
    policy.q_network.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()})
    if not bool(g["vdn"]):
        trainer.mixer.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, "mixer/").items()})
    trainer.hard_target_updates()
    return dims, buf, policy, trainer


@pytest.mark.parametrize("name", CASES)
def test_buffer_and_train_steps_match_reference(name):
    g = load_golden(name)
    dims, buf, policy, trainer = build(g)
    s = buf.policy_buffers["policy_0"].sample_inds(g["inds"])
    for k, a in zip(T_KEYS, s):
        assert tuple(a.shape) == g["batch/" + k].shape, k
        assert np.array_equal(a.cpu().numpy(), g["batch/" + k]), k          # gather is bit-exact
    w = g["per_weights"] if "per_weights" in g else None
    batch = tuple({"policy_0": a} for a in s) + (w, g["inds"] if w is not None else None)
    for st in range(len(g["loss"])):
        info, prio, _ = trainer.train_policy_on_batch(batch, True)
        trainer.soft_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["Q_tot"]), g["Q_tot"][st], rtol=RTOL, atol=1e-6)
        if w is not None:
            np.testing.assert_allclose(prio, g["priorities"][st], rtol=RTOL, atol=1e-6)
    live = dict(policy.q_network.named_parameters())
    tgt = dict(trainer.target_policies["policy_0"].q_network.named_parameters())
    for k, ref in sub(g, "final_agent/").items():
        np.testing.assert_allclose(live[k].detach().cpu().numpy(), ref, rtol=0, atol=3e-5, err_msg=k)
    for k, ref in sub(g, "final_agent_tgt/").items():
        np.testing.assert_allclose(tgt[k].detach().cpu().numpy(), ref, rtol=0, atol=3e-5, err_msg="tgt " + k)
    if not bool(g["vdn"]):
        for k, ref in sub(g, "final_mixer/").items():
            np.testing.assert_allclose(dict(trainer.mixer.named_parameters())[k].detach().cpu().numpy(), ref, rtol=0, atol=3e-5, err_msg=k)


def test_numpy_batch_from_reference_buffer_is_accepted():
    """The trainer also takes the reference buffer's numpy 13-tuple (host arrays are uploaded and stacked)."""
    g = load_golden("mqmix_spread")
    dims, buf, policy, trainer = build(g)
    batch = tuple({"policy_0": g["batch/" + k]} for k in T_KEYS) + (None, None)
    info, _, _ = trainer.train_policy_on_batch(batch, True)
    np.testing.assert_allclose(float(info["loss"]), g["loss"][0], rtol=RTOL)


@pytest.mark.parametrize("name", ["mqmix_spread", "mqmix_shape_nofn", "mqmix_shape_tanh", "mqmix_var_tanh_nofn_huber_per"])
def test_policy_q_values_match_oracle(name):
    from oracle import mqmix_oracle as MO
    g = load_golden(name)
    dims, buf, policy, trainer = build(g)
    assert list(policy.q_network.state_dict().keys()) == list(sub(g, "agent/").keys())      # (no feature_norm.* without the input LayerNorm)
    P = {k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()}
    torch.manual_seed(0)
    x = torch.randn(37, dims.obs_dim)
    ref = MO.mlp_agent_q(P, x, use_relu=bool(g["hp_use_relu"]) if "hp_use_relu" in g else True)
    got = policy.get_q_values(x.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6)
    acts, gq = policy.get_actions(x.cuda())
    assert np.array_equal(acts.argmax(-1), ref.argmax(-1).numpy())


def test_per_agent_centralized_observations_use_agent_zero():
    """use_same_share_obs = False (mlp_buffer.py:135-139, mqmix.py:78-84): the buffer keeps one centralized observation per
    agent and returns it as [N, B, S]; the mixer reads agent 0's. So a per-agent run whose agent-0 copies equal the shared
    observations of the reference fixture -- and whose other agents' copies are garbage -- must reproduce the fixture."""
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    from offpolicy_amd.utils.synth import policy_info_for
    g = load_golden("mqmix_spread")
    dims, _, policy, trainer = build(g)
    n = dims.n_agents
    buf = MlpReplayBuffer(policy_info_for(dims), {"policy_0": list(range(n))}, int(g["cap"]), False, True, False, device="cuda:0")
    rng = np.random.RandomState(1)

    def per_agent(x):           # [n_steps, S] -> [n_steps, N, S]: agent 0 = the shared observation, the rest noise
        out = rng.standard_normal((x.shape[0], n, x.shape[-1])).astype(np.float32)
        out[:, 0] = x if x.ndim == 2 else x[:, 0]
        return out
    for pre in (["pre_tr/"] if "pre_idx_range" in g else []) + ["tr/"]:
        d = {k: g[pre + k] for k in T_KEYS}
        d["share_obs"], d["next_share_obs"] = per_agent(d["share_obs"]), per_agent(d["next_share_obs"])
        buf.insert(d["obs"].shape[0], *[{"policy_0": d[k]} for k in T_KEYS])
    s = buf.policy_buffers["policy_0"].sample_inds(g["inds"])
    B = len(g["inds"])
    assert tuple(s[1].shape) == (n, B, dims.state_dim) and tuple(s[5].shape) == (n, B, dims.state_dim)
    ref_share = g["batch/share_obs"] if g["batch/share_obs"].ndim == 2 else g["batch/share_obs"][:, 0]
    assert np.array_equal(s[1][0].cpu().numpy(), ref_share)
    batch = tuple({"policy_0": a} for a in s) + (None, None)
    for st in range(len(g["loss"])):
        info, _, _ = trainer.train_policy_on_batch(batch, False)
        trainer.soft_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][st], rtol=RTOL)


@pytest.mark.parametrize("name", ["mqmix_md_small", "mqmix_md_huber_per"])
def test_multi_discrete_train_steps_match_reference(name):
    """Round 5: MultiDiscrete action spaces under MLP QMIX on the engine (mqmix.py:41-51, 116-130, 144-155; mQMixPolicy.py:47-55): one q head per
    sub-action, one mixer input per (agent, sub-action) -- run by the Discrete kernels with every (agent, sub-action) pair presented as an
    agent whose availability mask is the sub-action's block (QMix._md_expand). The reference buffer's numpy 13-tuple (no availability
    masks) through M_QMix: losses, gradient norms, Q_tot, priorities and the parameters after the fixture's steps."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims, policy_info_for
    from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy
    from offpolicy_amd.algorithms.mqmix.mqmix import M_QMix
    g = load_golden(name)
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    heads = [int(x) for x in g["multi_discrete"]]
    dims = EnvDims("fx", n, a, d, s, 1)
    args = default_args(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]),
                        use_huber_loss=bool(g["hp_huber"]), huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]),
                        per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]), tau=float(g["hp_tau"]),
                        max_grad_norm=float(g["hp_maxnorm"]), use_double_q=bool(g["hp_double_q"]))
    dev = torch.device("cuda:0")
    pinfo = policy_info_for(dims, multi_discrete=heads)
    policy = M_QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
    assert policy.multidiscrete and policy.head_dims == heads and policy.output_dim == a
    trainer = M_QMix(args, n, {"policy_0": policy}, lambda x: "policy_0", device=dev)
    assert trainer.num_mixer_q_inps == n * len(heads)
    assert list(policy.q_network.state_dict().keys()) == list(sub(g, "agent/").keys())      # upstream's per-head names, in order
    policy.q_network.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()})
    trainer.mixer.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, "mixer/").items()})
    trainer.hard_target_updates()
    w = g["per_weights"] if "per_weights" in g else None
    batch = tuple(({"policy_0": g["batch/" + k]} if "batch/" + k in g else None) for k in T_KEYS) + (w, g["inds"] if w is not None else None)
    for st in range(len(g["loss"])):
        info, prio, _ = trainer.train_policy_on_batch(batch, True)
        trainer.soft_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["Q_tot"]), g["Q_tot"][st], rtol=RTOL, atol=1e-6)
        if w is not None:
            np.testing.assert_allclose(prio, g["priorities"][st], rtol=RTOL, atol=1e-6)
    live = dict(policy.q_network.named_parameters())
    tgt = dict(trainer.target_policies["policy_0"].q_network.named_parameters())
    for k, ref in sub(g, "final_agent/").items():
        np.testing.assert_allclose(live[k].detach().cpu().numpy(), ref, rtol=0, atol=3e-5, err_msg=k)
    for k, ref in sub(g, "final_agent_tgt/").items():
        np.testing.assert_allclose(tgt[k].detach().cpu().numpy(), ref, rtol=0, atol=3e-5, err_msg="tgt " + k)
    for k, ref in sub(g, "final_mixer/").items():
        np.testing.assert_allclose(dict(trainer.mixer.named_parameters())[k].detach().cpu().numpy(), ref, rtol=0, atol=3e-5, err_msg=k)
