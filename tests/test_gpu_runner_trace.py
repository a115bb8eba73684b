"""-m gpu: replay of what the REFERENCE's SMACRunner did to its buffer / policy / trainer (tests/golden/runner_trace_qmix.npz,
recorded by oracle/make_runner_trace.py from the real runner -- constructor, warm-up with random actions, three run() cycles
of epsilon-greedy collection + insert + sample + train_policy_on_batch + soft update, save_q) against the ENGINE's classes:
same call order, same arguments (the numpy arrays the runner built), same numpy / torch RNG states, every return value
compared with what the reference's own classes returned to the runner. (the runner drives the engine.)"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _set_rng(c):
    np.random.set_state(("MT19937", c["in/np_keys"], int(c["in/np_pos"][0]), int(c["in/np_pos"][1]), float(c["in/np_gauss"][0])))
    torch.set_rng_state(torch.from_numpy(c["in/torch"]))


def _call(g, i):
    pre = "c%03d/" % i
    return {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}


def test_reference_runner_call_trace_replays_on_the_engine():
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.spaces import Discrete
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    g = load_golden("runner_trace_qmix")
    calls = [str(x) for x in g["calls"]]
    N, A, D, S, T = [int(x) for x in g["dims"]]
    batch_size, buffer_size, lr, eps0, eps1, eps_t = g["hp"]
    args = default_args(batch_size=int(batch_size), buffer_size=int(buffer_size), lr=float(lr), epsilon_start=float(eps0),
                        epsilon_finish=float(eps1), epsilon_anneal_time=float(eps_t), episode_length=T)
    dev = torch.device("cuda:0")
    pinfo = {"policy_0": {"cent_obs_dim": S, "cent_act_dim": A * N, "obs_space": [D], "share_obs_space": [S], "act_space": Discrete(A)}}
    policy = trainer = buf = None
    last_sample = None
    seen = {k: 0 for k in set(calls)}
    for i, name in enumerate(calls):
        c = _call(g, i)
        seen[name] += 1
        if name == "policy.__init__":
            _set_rng(c)
            policy = QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
            sd = {k[len("out/sd/"):]: v for k, v in c.items() if k.startswith("out/sd/")}
            ours = policy.q_network.state_dict()
            assert list(ours.keys()) == list(sd.keys())
            for k, v in sd.items():      # same RNG stream -> same initial weights (to LAPACK-QR rounding across hosts)
                np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=3e-5, err_msg=k)
            policy.q_network.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
            assert policy.output_dim == A and policy.obs_dim == D and policy.central_obs_dim == S     # attributes the runner reads
        elif name == "trainer.__init__":
            _set_rng(c)
            trainer = QMix(args, N, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=T)
            sd = {k[len("out/sd/"):]: v for k, v in c.items() if k.startswith("out/sd/")}
            ours = trainer.mixer.state_dict()
            assert list(ours.keys()) == list(sd.keys())
            for k, v in sd.items():
                np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=3e-5, err_msg=k)
            trainer.mixer.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
            trainer.hard_target_updates()       # targets = copies of the (just loaded) live networks, as at construction
            buf = RecReplayBuffer(pinfo, {"policy_0": list(range(N))}, int(buffer_size), T, True, True, False, device=dev)
        elif name == "trainer.prep_rollout":
            trainer.prep_rollout()
        elif name == "trainer.prep_training":
            trainer.prep_training()
        elif name == "policy.get_random_actions":
            _set_rng(c)
            acts = policy.get_random_actions(c["in/obs"], c["in/available_actions"])
            assert np.array_equal(np.asarray(acts), c["out/actions"]), i
        elif name == "policy.get_actions":
            _set_rng(c)
            explore = bool(c["in/explore"])
            t_env = int(c["in/t_env"]) if "in/t_env" in c else None
            acts, h, gq = policy.get_actions(c["in/obs"], c["in/prev_actions"], c["in/rnn_states"], c["in/available_actions"],
                                             t_env=t_env, explore=explore)
            np.testing.assert_allclose(h.cpu().numpy() if torch.is_tensor(h) else h, c["out/rnn_states"], rtol=1e-4, atol=2e-5, err_msg=str(i))
            gq = gq.cpu().numpy() if torch.is_tensor(gq) else np.asarray(gq)
            assert gq.shape == c["out/greedy_Qs"].shape, (i, gq.shape, c["out/greedy_Qs"].shape)
            np.testing.assert_allclose(gq, c["out/greedy_Qs"], rtol=1e-3, atol=3e-5, err_msg=str(i))
            acts = np.asarray(acts)
            assert acts.shape == c["out/actions"].shape
            if not np.array_equal(acts, c["out/actions"]):
                # only a near-tie between two Q values may flip a greedy action
                q, _ = policy.get_q_values(c["in/obs"], c["in/prev_actions"], c["in/rnn_states"])
                q = q.cpu().numpy()
                q[c["in/available_actions"] == 0] = -1e10
                for r in np.nonzero((acts != c["out/actions"]).any(-1))[0]:
                    top = np.sort(q[r])[-2:]
                    assert top[1] - top[0] < 1e-4, (i, r, q[r])
        elif name == "buffer.insert":
            d = {k: {"policy_0": c["in/" + k]} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")}
            idx = buf.insert(int(c["in/n"]), d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
            assert np.array_equal(np.asarray(idx), c["out/idx_range"]), i
        elif name == "buffer.sample":
            _set_rng(c)
            last_sample = buf.sample(int(c["in/batch_size"]))
            for j, k in enumerate(("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")):
                got = last_sample[j]["policy_0"]
                assert tuple(got.shape) == c["out/" + k].shape, (k, got.shape)
                assert np.array_equal(got.cpu().numpy(), c["out/" + k]), (i, k)      # the HIP gather returns the reference's batch
            assert last_sample[7] is None and last_sample[8] is None
        elif name == "trainer.train_policy_on_batch":
            info, prio, idxes = trainer.train_policy_on_batch(last_sample)
            assert prio is None and idxes is None
            np.testing.assert_allclose(float(info["loss"]), float(c["out/loss"]), rtol=2e-3, err_msg=str(i))
            np.testing.assert_allclose(float(info["grad_norm"]), float(c["out/grad_norm"]), rtol=2e-3, err_msg=str(i))
            np.testing.assert_allclose(float(info["Q_tot"]), float(c["out/Q_tot"]), rtol=2e-3, atol=1e-5, err_msg=str(i))
        elif name == "trainer.soft_target_updates":
            trainer.soft_target_updates()
        elif name == "runner.save_q":
            q_sd, m_sd = policy.q_network.state_dict(), trainer.mixer.state_dict()
            ref_q = {k[len("out/q/"):]: v for k, v in c.items() if k.startswith("out/q/")}
            ref_m = {k[len("out/m/"):]: v for k, v in c.items() if k.startswith("out/m/")}
            assert list(q_sd.keys()) == list(ref_q.keys()) and list(m_sd.keys()) == list(ref_m.keys())
            for ours, ref in ((q_sd, ref_q), (m_sd, ref_m)):
                for k, v in ref.items():
                    np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=1e-4, err_msg=k)
        else:
            raise AssertionError("unknown call in the trace: " + name)
    assert seen["policy.get_actions"] > 20 and seen["buffer.insert"] >= 8 and seen["trainer.train_policy_on_batch"] == 3


@pytest.mark.parametrize("trace", ["runner_trace_rmaddpg_multi", "runner_trace_rmaddpg_sl"])
def test_reference_mpe_runner_multi_policy_trace_replays_on_the_engine(trace):
    """The reference's MPERunner with ONE POLICY PER AGENT (share_policy = False, scripts/train_mpe_rmaddpg.sh), recorded by
    oracle/make_runner_trace_mpe.py (tests/golden/runner_trace_rmaddpg_multi.npz): three R_MADDPGPolicy constructions, the
    R_MADDPG trainer, warm-up through `separated_collect_rollout` (every agent queried through its own policy: random actions +
    a recurrent state update), three run() cycles (exploring rollout -> three-policy buffer.insert -> per policy buffer.sample ->
    shared_train_policy_on_batch -> soft updates of every policy) -- 203 calls, replayed against the engine's classes with the
    recorded arguments and RNG states; every return value is compared, and at the end all 12 networks.
    `runner_trace_rmaddpg_sl` (OPE_TRACE=sl): the same life cycle with the shapes of simple_speaker_listener, the scenario that script names --
    two agents, 3 / 11 observations, 3 / 5 actions: policies that differ in observation AND action width (151 calls, 8 networks)."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.spaces import Discrete
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.r_maddpg.algorithm.rMADDPGPolicy import R_MADDPGPolicy
    from offpolicy_amd.algorithms.r_maddpg.r_maddpg import R_MADDPG
    g = load_golden(trace)
    calls = [str(x) for x in g["calls"]]
    N, A, D, S, T = [int(x) for x in g["dims"]]
    As = [int(x) for x in g["act_dims"]] if "act_dims" in g else [A] * N
    Ds = [int(x) for x in g["obs_dims"]] if "obs_dims" in g else [D] * N
    batch_size, buffer_size, lr, eps0, eps1, eps_t = g["hp"]
    args = default_args(batch_size=int(batch_size), buffer_size=int(buffer_size), lr=float(lr), epsilon_start=float(eps0),
                        epsilon_finish=float(eps1), epsilon_anneal_time=float(eps_t), episode_length=T)
    dev = torch.device("cuda:0")
    pids = ["policy_%d" % i for i in range(N)]
    pinfo = {p: {"cent_obs_dim": S, "cent_act_dim": sum(As), "obs_space": [d], "share_obs_space": [S], "act_space": Discrete(a)}
             for p, d, a in zip(pids, Ds, As)}
    keys = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env")
    policies, trainer, buf, last_sample = {}, None, None, None
    seen = {k: 0 for k in set(calls)}
    to_np = lambda x: x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    for i, name in enumerate(calls):
        c = _call(g, i)
        seen[name] += 1
        pid = pids[int(c["in/pid"])] if "in/pid" in c else None
        if name == "policy.__init__":
            _set_rng(c)
            pol = R_MADDPGPolicy({"args": args, "device": dev}, pinfo[pid])
            for grp, mod in (("actor", pol.actor), ("critic", pol.critic)):
                sd = {k[len("out/sd/%s/" % grp):]: v for k, v in c.items() if k.startswith("out/sd/%s/" % grp)}
                ours = mod.state_dict()
                assert list(ours.keys()) == list(sd.keys())
                for k, v in sd.items():      # same RNG stream -> same initial weights (to LAPACK-QR rounding across hosts)
                    np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=3e-5, err_msg="%s %s %s" % (pid, grp, k))
                mod.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
            pol.hard_target_updates()
            assert pol.output_dim == As[pids.index(pid)] and pol.obs_dim == Ds[pids.index(pid)] and pol.central_obs_dim == S and pol.hidden_size == 64
            policies[pid] = pol
        elif name == "trainer.__init__":
            trainer = R_MADDPG(args, N, policies, lambda a: "policy_%d" % a, device=dev, episode_length=T)
            buf = RecReplayBuffer(pinfo, {p: [j] for j, p in enumerate(pids)}, int(buffer_size), T, True, False, False, device=dev)
        elif name == "trainer.prep_rollout":
            trainer.prep_rollout()
        elif name == "trainer.prep_training":
            trainer.prep_training()
        elif name == "policy.get_random_actions":
            _set_rng(c)
            assert np.array_equal(np.asarray(policies[pid].get_random_actions(c["in/obs"])), c["out/actions"]), i
        elif name == "policy.get_actions":
            _set_rng(c)
            explore = bool(c["in/explore"])
            t_env = int(c["in/t_env"]) if "in/t_env" in c else None
            acts, h, _ = policies[pid].get_actions(c["in/obs"], c["in/prev_actions"], c["in/rnn_states"], t_env=t_env, explore=explore)
            np.testing.assert_allclose(to_np(h).reshape(c["out/rnn_states"].shape), c["out/rnn_states"], rtol=1e-4, atol=2e-5, err_msg=str(i))
            acts = to_np(acts)
            assert acts.shape == c["out/actions"].shape, (i, acts.shape)
            if not np.array_equal(acts, c["out/actions"]):
                # only a near-tie between the two largest (noisy) logits may flip a one-hot action
                lg, _ = policies[pid].actor(c["in/obs"], c["in/prev_actions"], c["in/rnn_states"])
                lg = to_np(lg).reshape(acts.shape)
                for r in np.nonzero((acts != c["out/actions"]).any(-1))[0]:
                    a0, a1 = int(acts[r].argmax()), int(c["out/actions"][r].argmax())
                    assert explore or abs(lg[r, a0] - lg[r, a1]) < 1e-4, (i, r, lg[r])
                assert (acts != c["out/actions"]).any(-1).mean() < 0.05
        elif name == "buffer.insert":
            d = {k: {p: c["in/%s/%s" % (p, k)] for p in pids} for k in keys}
            idx = buf.insert(int(c["in/n"]), d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], {p: None for p in pids})
            assert np.array_equal(np.asarray(idx), c["out/idx_range"]), i
        elif name == "buffer.sample":
            _set_rng(c)
            last_sample = buf.sample(int(c["in/batch_size"]))
            for p in pids:
                for j, k in enumerate(keys):
                    got = last_sample[j][p]
                    assert tuple(got.shape) == c["out/%s/%s" % (p, k)].shape, (p, k, got.shape)
                    assert np.array_equal(got.cpu().numpy(), c["out/%s/%s" % (p, k)]), (i, p, k)
        elif name == "trainer.shared_train_policy_on_batch":
            _set_rng(c)
            info, prio, _ = trainer.shared_train_policy_on_batch(pid, last_sample)
            assert prio is None and bool(info["update_actor"])
            for k in ("critic_loss", "critic_grad_norm", "actor_loss", "actor_grad_norm"):
                np.testing.assert_allclose(float(info[k]), float(c["out/" + k]), rtol=2e-3, atol=1e-5, err_msg="%d %s" % (i, k))
        elif name == "policy.soft_target_updates":
            policies[pid].soft_target_updates()
        elif name == "runner.final_state":
            for p in pids:
                for grp, mod in (("actor", policies[p].actor), ("critic", policies[p].critic), ("target_actor", policies[p].target_actor),
                                 ("target_critic", policies[p].target_critic)):
                    pre = "out/%s/%s/" % (p, grp)
                    ref = {k[len(pre):]: v for k, v in c.items() if k.startswith(pre)}
                    ours = mod.state_dict()
                    assert list(ours.keys()) == list(ref.keys())
                    for k, v in ref.items():
                        np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=1e-4, err_msg="%s %s %s" % (p, grp, k))
        else:
            raise AssertionError("unknown call in the trace: " + name)
    assert seen["policy.get_actions"] > 60 and seen["buffer.insert"] >= 8 and seen["trainer.shared_train_policy_on_batch"] == 3 * N


def test_reference_mlp_runner_maddpg_trace_replays_on_the_engine():
    """The reference's MLP MPERunner (offpolicy/runner/mlp/mpe_runner.py; scripts/train_mpe_maddpg.sh: MADDPG, one shared policy),
    recorded by oracle/make_runner_trace_mlp.py (tests/golden/runner_trace_maddpg.npz): MADDPGPolicy + MADDPG + MlpReplayBuffer
    construction, warm-up with random actions, two run() cycles -- per environment step get_actions(explore) -> 12-argument
    buffer.insert -> every `train_interval` steps buffer.sample -> shared_train_policy_on_batch -> soft_target_updates. 76 calls
    replayed against the engine's classes with the recorded arguments and RNG states; every return value is compared, and at the end
    all four networks."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.spaces import Discrete
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy
    from offpolicy_amd.algorithms.maddpg.maddpg import MADDPG
    g = load_golden("runner_trace_maddpg")
    calls = [str(x) for x in g["calls"]]
    N, A, D, S, T = [int(x) for x in g["dims"]]
    batch_size, buffer_size, lr, eps0, eps1, eps_t = g["hp"]
    args = default_args(batch_size=int(batch_size), buffer_size=int(buffer_size), lr=float(lr), epsilon_start=float(eps0),
                        epsilon_finish=float(eps1), epsilon_anneal_time=float(eps_t), episode_length=T)
    dev = torch.device("cuda:0")
    P = "policy_0"
    pinfo = {P: {"cent_obs_dim": S, "cent_act_dim": A * N, "obs_space": [D], "share_obs_space": [S], "act_space": Discrete(A)}}
    keys = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition")
    policy, trainer, buf, last_sample = None, None, None, None
    seen = {k: 0 for k in set(calls)}
    to_np = lambda x: x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    for i, name in enumerate(calls):
        c = _call(g, i)
        seen[name] += 1
        if name == "policy.__init__":
            _set_rng(c)
            policy = MADDPGPolicy({"args": args, "device": dev}, pinfo[P])
            for grp, mod in (("actor", policy.actor), ("critic", policy.critic)):
                sd = {k[len("out/sd/%s/" % grp):]: v for k, v in c.items() if k.startswith("out/sd/%s/" % grp)}
                ours = mod.state_dict()
                assert list(ours.keys()) == list(sd.keys())          # the critic's Q heads are absent, as upstream (A-4)
                for k, v in sd.items():      # same RNG stream -> same initial weights (to LAPACK-QR rounding across hosts)
                    np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=3e-5, err_msg="%s %s" % (grp, k))
                mod.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
            for crit, pre in ((policy.critic, "out/head/0/"), (policy.target_critic, "out/thead/0/")):       # unsynchronised heads (A-4)
                np.testing.assert_allclose(crit._head_w.cpu().numpy().reshape(-1), c[pre + "weight"].reshape(-1), rtol=0, atol=3e-5)
                crit._head_w.copy_(torch.as_tensor(c[pre + "weight"]).reshape(crit._head_w.shape))
                crit._head_b.copy_(torch.as_tensor(c[pre + "bias"]).reshape(crit._head_b.shape))
            policy.hard_target_updates()
            assert policy.output_dim == A and policy.obs_dim == D and policy.central_obs_dim == S
        elif name == "trainer.__init__":
            trainer = MADDPG(args, N, {P: policy}, lambda a: P, device=dev)
            buf = MlpReplayBuffer(pinfo, {P: list(range(N))}, int(buffer_size), True, False, False, device=dev)
        elif name == "trainer.prep_rollout":
            trainer.prep_rollout()
        elif name == "trainer.prep_training":
            trainer.prep_training()
        elif name == "policy.get_random_actions":
            _set_rng(c)
            assert np.array_equal(np.asarray(policy.get_random_actions(c["in/obs"])), c["out/actions"]), i
        elif name == "policy.get_actions":
            _set_rng(c)
            explore = bool(c["in/explore"])
            t_env = int(c["in/t_env"]) if "in/t_env" in c else None
            acts, _ = policy.get_actions(c["in/obs"], t_env=t_env, explore=explore)
            acts = to_np(acts)
            assert acts.shape == c["out/actions"].shape, (i, acts.shape)
            if not np.array_equal(acts, c["out/actions"]):
                # only a near-tie between the two largest noisy logits may flip a one-hot action
                lg = to_np(policy.actor(c["in/obs"])).reshape(acts.shape)
                for r in np.nonzero((acts != c["out/actions"]).any(-1))[0]:
                    a0, a1 = int(acts[r].argmax()), int(c["out/actions"][r].argmax())
                    assert explore or abs(lg[r, a0] - lg[r, a1]) < 1e-4, (i, r, lg[r])
                assert (acts != c["out/actions"]).any(-1).mean() < 0.05
        elif name == "buffer.insert":
            d = [{P: c["in/" + k]} for k in keys]
            idx = buf.insert(int(c["in/n"]), *d, {P: None}, {P: None})
            assert np.array_equal(np.asarray(idx), c["out/idx_range"]), i
        elif name == "buffer.sample":
            _set_rng(c)
            last_sample = buf.sample(int(c["in/batch_size"]))
            assert len(last_sample) == 13
            for j, k in enumerate(keys):
                got = last_sample[j][P]
                assert tuple(got.shape) == c["out/" + k].shape, (k, got.shape)
                assert np.array_equal(got.cpu().numpy(), c["out/" + k]), (i, k)
        elif name == "trainer.shared_train_policy_on_batch":
            _set_rng(c)
            info, prio, _ = trainer.shared_train_policy_on_batch(P, last_sample)
            assert prio is None and bool(info["update_actor"])
            for k in ("critic_loss", "critic_grad_norm", "actor_loss", "actor_grad_norm"):
                np.testing.assert_allclose(float(info[k]), float(c["out/" + k]), rtol=2e-3, atol=1e-5, err_msg="%d %s" % (i, k))
        elif name == "policy.soft_target_updates":
            policy.soft_target_updates()
        elif name == "runner.final_state":
            for grp, mod in (("actor", policy.actor), ("critic", policy.critic), ("target_actor", policy.target_actor),
                             ("target_critic", policy.target_critic)):
                pre = "out/%s/" % grp
                ref = {k[len(pre):]: v for k, v in c.items() if k.startswith(pre)}
                ours = mod.state_dict()
                assert list(ours.keys()) == list(ref.keys())
                for k, v in ref.items():
                    np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=1e-4, err_msg="%s %s" % (grp, k))
        else:
            raise AssertionError("unknown call in the trace: " + name)
    assert seen["policy.get_actions"] == 10 and seen["buffer.insert"] == 25 and seen["trainer.shared_train_policy_on_batch"] == 5


def test_reference_mlp_runner_mvdn_one_policy_per_agent_trace_replays_on_the_engine():
    """The reference's MLP MPERunner with ONE POLICY PER AGENT under a VDN sum (scripts/train_mpe_mqmix.sh: mvdn on a speaker / listener
    pair of different observation width and action count), recorded by oracle/make_runner_trace_mvdn.py
    (tests/golden/runner_trace_mvdn_multi.npz): two M_QMixPolicy constructions, the M_QMix(vdn) trainer, random warm-up through
    `separated_collect_rollout`, two run() cycles (every agent queried through its own policy -> two-policy 12-argument insert -> every
    second step, per policy: sample -> train_policy_on_batch over ALL policies' networks -> soft updates). 112 calls replayed against
    the engine's classes with the recorded arguments and RNG states; every return value is compared, and at the end all four networks."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.spaces import Discrete
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy
    from offpolicy_amd.algorithms.mqmix.mqmix import M_QMix
    g = load_golden("runner_trace_mvdn_multi")
    calls = [str(x) for x in g["calls"]]
    shapes = [tuple(int(x) for x in r) for r in g["shapes"]]
    N, S, T = [int(x) for x in g["dims"]]
    batch_size, buffer_size, lr, eps0, eps1, eps_t = g["hp"]
    args = default_args(batch_size=int(batch_size), buffer_size=int(buffer_size), lr=float(lr), epsilon_start=float(eps0),
                        epsilon_finish=float(eps1), epsilon_anneal_time=float(eps_t), episode_length=T)
    dev = torch.device("cuda:0")
    pids = ["policy_%d" % i for i in range(N)]
    pinfo = {p: {"cent_obs_dim": S, "cent_act_dim": sum(a for _, a in shapes), "obs_space": [d], "share_obs_space": [S], "act_space": Discrete(a)}
             for p, (d, a) in zip(pids, shapes)}
    keys = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition")
    policies, trainer, buf, last_sample = {}, None, None, None
    seen = {k: 0 for k in set(calls)}
    for i, name in enumerate(calls):
        c = _call(g, i)
        seen[name] += 1
        pid = pids[int(c["in/pid"])] if "in/pid" in c else None
        if name == "policy.__init__":
            _set_rng(c)
            pol = M_QMixPolicy({"args": args, "device": dev}, pinfo[pid])
            sd = {k[len("out/sd/"):]: v for k, v in c.items() if k.startswith("out/sd/")}
            ours = pol.q_network.state_dict()
            assert list(ours.keys()) == list(sd.keys())
            for k, v in sd.items():      # same RNG stream -> same initial weights (to LAPACK-QR rounding across hosts)
                np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=3e-5, err_msg="%s %s" % (pid, k))
            pol.q_network.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
            policies[pid] = pol
        elif name == "trainer.__init__":
            trainer = M_QMix(args, N, policies, lambda a: "policy_%d" % a, device=dev, vdn=True)
            assert trainer.multi and trainer.vdn
            buf = MlpReplayBuffer(pinfo, {p: [j] for j, p in enumerate(pids)}, int(buffer_size), True, False, False, device=dev)
        elif name == "trainer.prep_rollout":
            trainer.prep_rollout()
        elif name == "trainer.prep_training":
            trainer.prep_training()
        elif name == "policy.get_random_actions":
            _set_rng(c)
            assert np.array_equal(np.asarray(policies[pid].get_random_actions(c["in/obs"])), c["out/actions"]), i
        elif name == "policy.get_actions":
            _set_rng(c)
            explore = bool(c["in/explore"])
            t_env = int(c["in/t_env"]) if "in/t_env" in c else None
            acts, _ = policies[pid].get_actions(c["in/obs"], t_env=t_env, explore=explore)
            acts = np.asarray(acts)
            assert acts.shape == c["out/actions"].shape, (i, acts.shape)
            if not np.array_equal(acts, c["out/actions"]):
                # only a near-tie between the two largest q values may flip a greedy action
                q = policies[pid].get_q_values(c["in/obs"]).detach().cpu().numpy().reshape(acts.shape)
                for r in np.nonzero((acts != c["out/actions"]).any(-1))[0]:
                    a0, a1 = int(acts[r].argmax()), int(c["out/actions"][r].argmax())
                    assert abs(q[r, a0] - q[r, a1]) < 1e-4, (i, r, q[r])
        elif name == "buffer.insert":
            d = [{p: c["in/%s/%s" % (p, k)] for p in pids} for k in keys]
            idx = buf.insert(int(c["in/n"]), *d, {p: None for p in pids}, {p: None for p in pids})
            assert np.array_equal(np.asarray(idx), c["out/idx_range"]), i
        elif name == "buffer.sample":
            _set_rng(c)
            last_sample = buf.sample(int(c["in/batch_size"]))
            assert len(last_sample) == 13
            for p in pids:
                for j, k in enumerate(keys):
                    got = last_sample[j][p]
                    assert tuple(got.shape) == c["out/%s/%s" % (p, k)].shape, (p, k, got.shape)
                    assert np.array_equal(got.cpu().numpy(), c["out/%s/%s" % (p, k)]), (i, p, k)
        elif name == "trainer.train_policy_on_batch":
            _set_rng(c)
            info, prio, _ = trainer.train_policy_on_batch(last_sample, bool(c["in/use_same_share_obs"]))
            assert prio is None
            for k in ("loss", "grad_norm", "Q_tot"):
                np.testing.assert_allclose(float(info[k]), float(c["out/" + k]), rtol=2e-3, atol=1e-5, err_msg="%d %s" % (i, k))
        elif name == "trainer.soft_target_updates":
            trainer.soft_target_updates()
        elif name == "runner.final_state":
            for p in pids:
                for grp, mod in (("live", policies[p].q_network), ("target", trainer.target_policies[p].q_network)):
                    pre = "out/%s/%s/" % (p, grp)
                    ref = {k[len(pre):]: v for k, v in c.items() if k.startswith(pre)}
                    ours = mod.state_dict()
                    assert list(ours.keys()) == list(ref.keys())
                    for k, v in ref.items():
                        np.testing.assert_allclose(ours[k].cpu().numpy(), v, rtol=0, atol=1e-4, err_msg="%s %s %s" % (p, grp, k))
        else:
            raise AssertionError("unknown call in the trace: " + name)
    assert seen["policy.get_actions"] == 20 and seen["buffer.insert"] == 25 and seen["trainer.train_policy_on_batch"] == 10
