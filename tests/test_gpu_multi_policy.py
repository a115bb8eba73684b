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
            for k, ref in refs.items():
                tol = 2e-3 * max(np.abs(ref).max(), 1e-6)
                np.testing.assert_allclose(got[k], ref, rtol=0, atol=tol, err_msg="grad " + k)
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
