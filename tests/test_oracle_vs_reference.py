"""Live pin of the oracle and of our host-side logic against the REAL reference (only where /root/reference exists,
i.e. the build container; skipped on the GPU box)."""
import os

import numpy as np
import pytest
import torch

from oracle.ref_import import reference_available, load_reference, reference_args

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not mounted")


def _build_reference(dims, argv=()):
    load_reference()
    from gym.spaces import Discrete
    from offpolicy.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy.algorithms.qmix.qmix import QMix
    from offpolicy.utils.rec_buffer import RecReplayBuffer
    args = reference_args(argv)
    torch.manual_seed(7)
    np.random.seed(7)
    pinfo = {"policy_0": {"cent_obs_dim": dims.state_dim, "cent_act_dim": dims.act_dim * dims.n_agents,
                          "obs_space": [dims.obs_dim], "share_obs_space": [dims.state_dim], "act_space": Discrete(dims.act_dim)}}
    policy = QMixPolicy({"args": args, "device": torch.device("cpu")}, pinfo["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=torch.device("cpu"),
                   episode_length=dims.episode_length)
    buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, 7, dims.episode_length, True, True, False)
    return args, policy, trainer, buf


@pytest.mark.parametrize("argv", [(), ("--use_huber_loss", "--huber_delta", "0.5"), ("--prev_act_inp",)])
def test_oracle_tracks_reference_on_fresh_random_problem(argv):
    from oracle import qmix_oracle as O
    from offpolicy_amd.utils.synth import EnvDims, synth_episodes, as_policy_dicts
    dims = EnvDims("fresh", 3, 6, 20, 14, 9)
    args, policy, trainer, buf = _build_reference(dims, argv)
    rng = np.random.RandomState(11)
    for n in (5, 4):                 # second insert wraps the 7-slot ring
        ep = synth_episodes(rng, n, dims, avail="bernoulli", runner_padding=True)
        d = as_policy_dicts(ep)
        buf.insert(n, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    pb = buf.policy_buffers["policy_0"]
    store = dict(obs=pb.obs, share_obs=pb.share_obs, acts=pb.acts, rewards=pb.rewards, dones=pb.dones, dones_env=pb.dones_env,
                 avail_acts=pb.avail_acts)
    inds = np.array([6, 0, 3, 3, 1])
    ref_batch = pb.sample_inds(inds)
    ora_batch = O.sample_inds(store, inds)
    for a, b in zip(ref_batch, ora_batch):
        assert np.array_equal(a, b)
    hp = O.HP(use_huber_loss=bool(args.use_huber_loss), huber_delta=float(args.huber_delta), prev_act_inp=bool(args.prev_act_inp))
    orc = O.QMixOracle({k: v.detach().numpy() for k, v in policy.q_network.named_parameters()},
                       {k: v.detach().numpy() for k, v in trainer.mixer.named_parameters()}, dims.n_agents, hp)
    for _ in range(2):
        batch = tuple({"policy_0": a} for a in ref_batch) + (None, None)
        info, _, _ = trainer.train_policy_on_batch(batch)
        trainer.soft_target_updates()
        out = orc.train_step(ora_batch)
        np.testing.assert_allclose(out["loss"], float(info["loss"].detach()), rtol=2e-5)
        np.testing.assert_allclose(out["grad_norm"], float(info["grad_norm"]), rtol=2e-5)
    for k, v in policy.q_network.named_parameters():
        np.testing.assert_allclose(orc.agent[k].numpy(), v.detach().numpy(), atol=2e-5, err_msg=k)
    for k, v in trainer.target_mixer.named_parameters():
        np.testing.assert_allclose(orc.mixer_tgt[k].numpy(), v.detach().numpy(), atol=2e-5, err_msg=k)


def test_segment_trees_and_per_sampling_match_reference_semantics():
    """Our SumSegmentTree/MinSegmentTree give the reference's answers (reduce range convention, prefix-sum descent)."""
    load_reference()
    from offpolicy.utils.segment_tree import SumSegmentTree as RS, MinSegmentTree as RM
    from offpolicy_amd.utils.segment_tree import SumSegmentTree, MinSegmentTree
    rng = np.random.RandomState(3)
    cap = 32
    rs, rm, s, m = RS(cap), RM(cap), SumSegmentTree(cap), MinSegmentTree(cap)
    for _ in range(20):
        idx = np.unique(rng.randint(0, cap, size=5))
        val = rng.rand(len(idx)) + 0.05
        rs[idx] = val
        rm[idx] = val
        s[idx] = val
        m[idx] = val
        np.testing.assert_allclose(s.sum(), rs.sum(), rtol=1e-12)
        np.testing.assert_allclose(s.sum(0, 17), rs.sum(0, 17), rtol=1e-12)
        assert m.min() == rm.min()
        mass = rng.rand(6) * rs.sum(0, cap - 1)
        assert np.array_equal(s.find_prefixsum_idx(mass), rs.find_prefixsum_idx(mass))
        np.testing.assert_allclose(s[idx], rs[idx], rtol=0)


def test_ring_bookkeeping_matches_reference_buffer():
    from offpolicy_amd.utils.ring import RingIndex
    from offpolicy_amd.utils.synth import EnvDims, synth_episodes, as_policy_dicts
    dims = EnvDims("r", 2, 3, 4, 5, 3)
    _, _, _, buf = _build_reference(dims)
    ring = RingIndex(7)
    rng = np.random.RandomState(0)
    for n in (3, 3, 2, 7, 1):
        ep = synth_episodes(rng, n, dims)
        d = as_policy_dicts(ep)
        want = buf.insert(n, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
        got = ring.next_slots(n)
        assert np.array_equal(got, want)
        assert ring.filled_i == buf.policy_buffers["policy_0"].filled_i
        assert ring.current_i == buf.policy_buffers["policy_0"].current_i


@pytest.mark.parametrize("workload,n_threads", [("3m", 1), ("3s5z", 8)])
def test_timed_oracle_path_runs_at_reference_speed(workload, n_threads):
    """bench.py's cpu_baseline times oracle.qmix_oracle under `reference_speed_ops()` + `fused_gru=True` because the
    GPU box has no /root/reference. That timed path must (a) give the values of the explicit-formula oracle and (b) run
    at the real reference's speed: >= 0.9x its steps/s here, interleaved best-of-5, same data, same indices -- at SMAC 3m with
    1 thread (the reference's default n_training_threads, config.py:17) and at 3s5z with 8 threads, the configuration and
    thread count the bench's `speedup_vs_cpu` divides by. B=32."""
    import time
    from oracle import qmix_oracle as O
    from offpolicy_amd.utils.synth import DIMS, synth_episodes, as_policy_dicts
    dims = DIMS[workload]
    n_ep, B = (64, 32) if workload == "3m" else (40, 32)
    load_reference()
    from gym.spaces import Discrete
    from offpolicy.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy.algorithms.qmix.qmix import QMix
    from offpolicy.utils.rec_buffer import RecReplayBuffer
    args = reference_args(())
    torch.manual_seed(7)
    np.random.seed(7)
    pinfo = {"policy_0": {"cent_obs_dim": dims.state_dim, "cent_act_dim": dims.act_dim * dims.n_agents,
                          "obs_space": [dims.obs_dim], "share_obs_space": [dims.state_dim], "act_space": Discrete(dims.act_dim)}}
    policy = QMixPolicy({"args": args, "device": torch.device("cpu")}, pinfo["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=torch.device("cpu"),
                   episode_length=dims.episode_length)
    buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, n_ep, dims.episode_length, True, True, False)
    d = as_policy_dicts(synth_episodes(np.random.RandomState(0), n_ep, dims, avail="bernoulli"))
    buf.insert(n_ep, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    pb = buf.policy_buffers["policy_0"]
    store = dict(obs=pb.obs, share_obs=pb.share_obs, acts=pb.acts, rewards=pb.rewards, dones=pb.dones, dones_env=pb.dones_env,
                 avail_acts=pb.avail_acts)
    P = ({k: v.detach().numpy().copy() for k, v in policy.q_network.named_parameters()},
         {k: v.detach().numpy().copy() for k, v in trainer.mixer.named_parameters()})
    slow, fast = O.QMixOracle(P[0], P[1], dims.n_agents, O.HP()), O.QMixOracle(P[0], P[1], dims.n_agents, O.HP())
    inds = np.random.RandomState(1).choice(n_ep, B)
    a = slow.train_step(O.sample_inds(store, inds))
    with O.reference_speed_ops():
        b = fast.train_step(O.sample_inds(store, inds), fused_gru=True)
    np.testing.assert_allclose(b["loss"], a["loss"], rtol=1e-5)
    np.testing.assert_allclose(b["grad_norm"], a["grad_norm"], rtol=1e-5)
    for k in slow.agent:
        np.testing.assert_allclose(fast.agent[k].numpy(), slow.agent[k].numpy(), atol=1e-6, err_msg=k)

    threads = torch.get_num_threads()
    torch.set_num_threads(min(n_threads, os.cpu_count() or 1))
    try:
        rng = np.random.RandomState(2)

        def ref_step():
            s = pb.sample_inds(rng.choice(n_ep, B))
            trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s) + (None, None))
            trainer.soft_target_updates()

        def orc_step():
            with O.reference_speed_ops():
                fast.train_step(O.sample_inds(store, rng.choice(n_ep, B)), fused_gru=True, soft_update=True)
        best = {"ref": 1e9, "orc": 1e9}
        ref_step(); orc_step()
        # best-of timing, interleaved; a shared container's noise only ever makes a sample slower, so the comparison is repeated (up to three
        # rounds, keeping every sample's minimum) before the bound is declared missed
        for attempt in range(3):
            for _ in range(5 if workload == "3m" else 3):
                for name, f in (("ref", ref_step), ("orc", orc_step)):
                    t0 = time.perf_counter()
                    f(); f()
                    best[name] = min(best[name], (time.perf_counter() - t0) / 2)
            if best["ref"] / best["orc"] >= 0.9:
                break
    finally:
        torch.set_num_threads(threads)
    assert best["ref"] / best["orc"] >= 0.9, "timed oracle path %.1f ms/step vs reference %.1f ms/step" % (1e3 * best["orc"], 1e3 * best["ref"])
