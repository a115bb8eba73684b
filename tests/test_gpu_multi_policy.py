"""-m gpu: QMIX / VDN (recurrent and MLP) with SEVERAL POLICIES under one mixer (share_policy = False) against the reference's
frozen outputs (tests/golden/*multi*.npz from oracle/make_golden_multi.py): the `for p_id in self.policy_ids` loops of
offpolicy/algorithms/qmix/qmix.py:100-150 and mqmix/mqmix.py:95-178, policies of different observation width / action count / agent
count (scripts/train_mpe_mqmix.sh: MVDN on simple_speaker_listener). Same tolerances as tests/test_gpu_qmix.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu
RTOL = 1e-4
GRAD_TOL = 2e-5      # of each tensor's max magnitude: ~10x the worst error measured on the GPU, 1.6e-6 (round 6; was a blanket 2e-3; profiles/r06_parity_errors.txt)
M_KEYS = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition",
          "avail_acts", "next_avail_acts")
R_KEYS = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")


def build(g, mlp, device="cuda:0"):
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.spaces import Discrete
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    from offpolicy_amd.algorithms.mqmix.mqmix import M_QMix
    shapes = [tuple(int(x) for x in r) for r in g["shapes"]]
    S, T, vdn = int(g["S"]), int(g["T"]), bool(g["vdn"])
    args = default_args(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
                        huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]),
                        tau=float(g["hp_tau"]), max_grad_norm=float(g["hp_maxnorm"]), use_double_q=bool(g["hp_double_q"]), episode_length=T)
    dev = torch.device(device)
    pids = ["policy_%d" % i for i in range(len(shapes))]
    cent_act = sum(n * a for n, a, _ in shapes)
    pinfo = {p: {"cent_obs_dim": S, "cent_act_dim": cent_act, "obs_space": [d], "share_obs_space": [S], "act_space": Discrete(a)}
             for p, (n, a, d) in zip(pids, shapes)}
    owner, k = {}, 0
    for p, (n, _, _) in zip(pids, shapes):
        for a in range(k, k + n):
            owner[a] = p
        k += n
    torch.manual_seed(1)
    np.random.seed(1)
    policies = {p: (M_QMixPolicy if mlp else QMixPolicy)({"args": args, "device": dev}, pinfo[p]) for p in pids}
    if mlp:
        trainer = M_QMix(args, k, policies, lambda a: owner[a], device=dev, vdn=vdn)
    else:
        trainer = QMix(args, k, policies, lambda a: owner[a], device=dev, episode_length=T, vdn=vdn)
    # same seed -> the reference's initial draws (to LAPACK-QR rounding); then the fixture's exact values for the stepping part
    for i, p in enumerate(pids):
        sd = sub(g, "p%d/agent/" % i)
        ours = policies[p].q_network.state_dict()
        assert list(ours.keys()) == list(sd.keys())
        for kk, v in sd.items():
            np.testing.assert_allclose(ours[kk].cpu().numpy(), v, rtol=0, atol=3e-5, err_msg="%s %s" % (p, kk))
        policies[p].q_network.load_state_dict({kk: torch.as_tensor(v) for kk, v in sd.items()})
    if not vdn:
        sd = sub(g, "mixer/")
        ours = trainer.mixer.state_dict()
        assert list(ours.keys()) == list(sd.keys())
        for kk, v in sd.items():
            np.testing.assert_allclose(ours[kk].cpu().numpy(), v, rtol=0, atol=3e-5, err_msg="mixer " + kk)
        trainer.mixer.load_state_dict({kk: torch.as_tensor(v) for kk, v in sd.items()})
    if dev.type == "cuda":
        trainer.hard_target_updates()
    keys = M_KEYS if mlp else R_KEYS
    batch = tuple({p: (g["p%d/batch/%s" % (i, kk)] if "p%d/batch/%s" % (i, kk) in g else None) for i, p in enumerate(pids)} for kk in keys)
    return pids, policies, trainer, batch


def named(trainer, pids, flat):
    out = {}
    for i, p in enumerate(pids):
        q = trainer.policies[p].q_network
        o = trainer._poff[p]
        for name, (shape, off) in q.spec().items():
            out["p%d/agent/%s" % (i, name)] = flat[o + off:o + off + int(np.prod(shape))].view(shape).cpu().numpy()
    if not trainer.vdn:
        for name, (shape, off) in trainer.mixer.spec().items():
            out["mixer/" + name] = flat[off:off + int(np.prod(shape))].view(shape).cpu().numpy()
    return out


@pytest.mark.parametrize("name,mlp", [("mqmix_multi", True), ("mqmix_multi_huber_per", True), ("mvdn_multi_speaker_listener", True),
                                      ("qmix_multi", False), ("qmix_multi_nodouble", False), ("vdn_multi", False)])
def test_several_policies_under_one_mixer_match_reference(name, mlp):
    g = load_golden(name)
    pids, policies, trainer, batch = build(g, mlp)
    assert trainer.multi
    w = g["per_weights"] if "per_weights" in g else None
    full = batch + (w, g["inds"] if w is not None else None)
    for s in range(len(g["loss"])):
        info, prio, _ = trainer.train_policy_on_batch(full, True) if mlp else trainer.train_policy_on_batch(full)
        if s == 0:
            cnt = float(trainer.grad[trainer.numel + 1])
            coef = min(1.0, float(g["hp_maxnorm"]) / (float(g["grad_norm"][0]) + 1e-6))
            got = named(trainer, pids, trainer.grad[:trainer.numel] * (coef / cnt))
            refs = sub(g, "grad0/")
            assert len(refs) > 10 * len(pids)
            from golden_util import assert_grads_close
            assert_grads_close("multi_policy:" + name, got, refs, GRAD_TOL, floor=1e-6)
            for k in got:
                if ".fc_h." in k:
                    assert not np.any(got[k]), k
        trainer.soft_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["Q_tot"]), g["Q_tot"][s], rtol=RTOL, atol=1e-6)
        if w is not None:
            np.testing.assert_allclose(np.asarray(prio), g["priorities"][s], rtol=RTOL, atol=1e-6)
    live, tgt = named(trainer, pids, trainer.theta), named(trainer, pids, trainer.theta_tgt)
    for i, p in enumerate(pids):
        for grp, src in (("agent/", live), ("agent_tgt/", tgt)):
            for k, ref in sub(g, "final/p%d/%s" % (i, grp)).items():
                np.testing.assert_allclose(src["p%d/agent/%s" % (i, k)], ref, rtol=0, atol=3e-5, err_msg="%s %s %s" % (p, grp, k))
        # the nn.Module views see the same memory (checkpoint / rollout path)
        sd = policies[p].q_network.state_dict()
        np.testing.assert_array_equal(sd["q.action_out.weight"].cpu().numpy(), live["p%d/agent/q.action_out.weight" % i])
        sdt = trainer.target_policies[p].q_network.state_dict()
        np.testing.assert_array_equal(sdt["q.action_out.weight"].cpu().numpy(), tgt["p%d/agent/q.action_out.weight" % i])
    for grp, src in (("mixer/", live), ("mixer_tgt/", tgt)):
        for k, ref in sub(g, "final/" + grp).items():
            np.testing.assert_allclose(src["mixer/" + k], ref, rtol=0, atol=3e-5, err_msg=grp + k)


@pytest.mark.parametrize("name", ["maddpg_multi", "matd3_multi_per", "maddpg_multi_sl", "matd3_multi_actdims", "matd3_multi_kinds", "maddpg_multi_md"])
def test_mlp_maddpg_several_policies_match_reference(name):
    """MLP MADDPG / MATD3 with several policies (share_policy = False; get_update_info maddpg.py:40-80): every policy's own actor,
    critic and batch, the joint target action assembled from all target actors (ope_ddpg_target_actions per policy), each policy
    updated in turn as runner/mlp/base_runner.py:196-217 does -- fixtures from oracle/make_golden_ddpg.py (OPE_GOLDEN_ONLY=multi)."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.spaces import Discrete, Box, MultiDiscrete
    from test_ddpg_oracle_golden import mlp_multi_kinds
    from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy
    from offpolicy_amd.algorithms.matd3.algorithm.MATD3Policy import MATD3Policy
    from offpolicy_amd.algorithms.maddpg.maddpg import MADDPG
    from offpolicy_amd.algorithms.matd3.matd3 import MATD3
    g = load_golden(name)
    groups, dims_obs, A, S, td3 = [int(x) for x in g["groups"]], [int(x) for x in g["dims_obs"]], int(g["A"]), int(g["S"]), bool(g["td3"])
    N = sum(groups)
    args = default_args(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
                        huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_eps=float(g["hp_per_eps"]), tau=float(g["hp_tau"]),
                        max_grad_norm=float(g["hp_maxnorm"]))
    dev = torch.device("cuda:0")
    pids = ["policy_%d" % i for i in range(len(groups))]
    # (`maddpg_multi_sl`, `matd3_multi_actdims`: the policies differ in their NUMBER OF ACTIONS too -- simple_speaker_listener's 3 and 5;
    # the joint action is then described in columns, ope_ddpg_cfg.joint_act_dim)
    As = [int(x) for x in g["act_dims"]] if "act_dims" in g else [A] * len(groups)
    # (`matd3_multi_kinds`, `maddpg_multi_md`: policies of different action-space KINDS -- discrete, multi-discrete, continuous -- under one trainer)
    space_of = lambda a, kd: (Discrete(a) if kd is None else Box(low=-np.ones(a, np.float32), high=np.ones(a, np.float32)) if kd == "cont"
                              else MultiDiscrete([[0, k - 1] for k in kd[1]]))
    pinfo = {p: {"cent_obs_dim": S, "cent_act_dim": sum(a * m for a, m in zip(As, groups)), "obs_space": [d], "share_obs_space": [S],
                 "act_space": space_of(a, kd)} for p, d, a, kd in zip(pids, dims_obs, As, mlp_multi_kinds(g))}
    owner, k = {}, 0
    for p, m in zip(pids, groups):
        for a in range(k, k + m):
            owner[a] = p
        k += m
    torch.manual_seed(1)
    np.random.seed(1)
    policies = {p: (MATD3Policy if td3 else MADDPGPolicy)({"args": args, "device": dev}, pinfo[p]) for p in pids}
    trainer = (MATD3 if td3 else MADDPG)(args, N, policies, lambda a: owner[a], device=dev)
    assert trainer.multi_policy
    for i, p in enumerate(pids):
        pol = policies[p]
        for grp, mod in (("actor/", pol.actor), ("critic/", pol.critic), ("actor_tgt/", pol.target_actor), ("critic_tgt/", pol.target_critic)):
            ref = sub(g, "p%d/%s" % (i, grp))
            got = {kk: v.detach().cpu().numpy() for kk, v in mod.named_parameters()}
            for kk, v in ref.items():       # same RNG stream -> the reference's initial draws (to LAPACK-QR rounding)
                np.testing.assert_allclose(got[kk], v, rtol=0, atol=3e-5, err_msg="%s %s%s" % (p, grp, kk))
            mod.load_state_dict({kk: torch.as_tensor(v) for kk, v in ref.items()})
        for crit, pre in ((pol.critic, "p%d/heads/" % i), (pol.target_critic, "p%d/heads_tgt/" % i)):
            np.testing.assert_allclose(crit._head_w.cpu().numpy(), g[pre + "w"], atol=3e-5)
            crit._head_w.copy_(torch.as_tensor(g[pre + "w"]))
            crit._head_b.copy_(torch.as_tensor(g[pre + "b"]))
    w = g["per_weights"] if "per_weights" in g else None
    batch = tuple({p: g["p%d/batch/%s" % (i, kk)] for i, p in enumerate(pids)} for kk in M_KEYS) + (w, g["inds"] if w is not None else None)
    steps = len(g["p0/critic_loss"])
    for s in range(steps):
        for i, p in enumerate(pids):
            torch.manual_seed(1000 + 10 * s + i)
            info, prio, _ = trainer.shared_train_policy_on_batch(p, batch)
            assert info["update_actor"]
            np.testing.assert_allclose(float(info["critic_loss"]), g["p%d/critic_loss" % i][s], rtol=RTOL, err_msg="%d %s" % (s, p))
            np.testing.assert_allclose(float(info["critic_grad_norm"]), g["p%d/critic_grad_norm" % i][s], rtol=RTOL, err_msg="%d %s" % (s, p))
            np.testing.assert_allclose(float(info["actor_loss"]), g["p%d/actor_loss" % i][s], rtol=5e-4, atol=2e-6, err_msg="%d %s" % (s, p))
            np.testing.assert_allclose(float(info["actor_grad_norm"]), g["p%d/actor_grad_norm" % i][s], rtol=5e-4, err_msg="%d %s" % (s, p))
            if w is not None:
                np.testing.assert_allclose(np.asarray(prio), g["p%d/priorities" % i][s], rtol=RTOL)
        for p in pids:
            policies[p].soft_target_updates()
    for i, p in enumerate(pids):
        pol = policies[p]
        for grp, mod in (("actor/", pol.actor), ("critic/", pol.critic), ("actor_tgt/", pol.target_actor), ("critic_tgt/", pol.target_critic)):
            got = {kk: v.detach().cpu().numpy() for kk, v in mod.named_parameters()}
            for kk, ref in sub(g, "final/p%d/%s" % (i, grp)).items():
                np.testing.assert_allclose(got[kk], ref, rtol=0, atol=3e-5, err_msg="%s final %s%s" % (p, grp, kk))
