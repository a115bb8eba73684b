"""-m gpu: MLP MADDPG / MATD3 through the C-ABI vs the reference's frozen outputs (same gumbel noise stream)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from test_mlp_oracle_golden import T_KEYS
from test_ddpg_oracle_golden import CASES

pytestmark = pytest.mark.gpu
RTOL = 2e-4
GRAD_TOL = 2e-5      # of each tensor's max magnitude: ~10x the worst error measured on the GPU, 1.5e-6 (config 3 at B = 256, both families; fused vs separate / graph vs eager: 2e-7) (round 6; was a blanket 2e-3; profiles/r06_parity_errors.txt)


def build(g, device="cuda:0", same_share=True):
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims, policy_info_for
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy
    from offpolicy_amd.algorithms.matd3.algorithm.MATD3Policy import MATD3Policy
    from offpolicy_amd.algorithms.maddpg.maddpg import MADDPG
    from offpolicy_amd.algorithms.matd3.matd3 import MATD3
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    dims = EnvDims("fx", n, a, d, s, 1)
    args = default_args(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
                        huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_eps=float(g["hp_per_eps"]), tau=float(g["hp_tau"]),
                        max_grad_norm=float(g["hp_maxnorm"]), weight_decay=float(g["hp_wd"]) if "hp_wd" in g else 0.0,
                        use_same_share_obs=same_share)
    pinfo = policy_info_for(dims, continuous=bool(g["continuous"]) if "continuous" in g else False,
                            multi_discrete=g["multi_discrete"] if "multi_discrete" in g else None)
    dev = torch.device(device)
    torch.manual_seed(1)
    np.random.seed(1)
    td3 = bool(g["td3"])
    policy = (MATD3Policy if td3 else MADDPGPolicy)({"args": args, "device": dev}, pinfo["policy_0"])
    trainer = (MATD3 if td3 else MADDPG)(args, n, {"policy_0": policy}, lambda x: "policy_0", device=dev)
    buf = MlpReplayBuffer(pinfo, {"policy_0": list(range(n))}, int(g["cap"]), same_share, True, False, device=device)
    buf.insert(len(g["idx_range"]), *[{"policy_0": g["tr/" + k]} for k in T_KEYS])
    return dims, buf, policy, trainer


def params_of(mod):
    return {k: v.detach().cpu().numpy() for k, v in mod.named_parameters()}


@pytest.mark.parametrize("name", CASES)
def test_construction_and_train_steps_match_reference(name):
    g = load_golden(name)
    dims, buf, policy, trainer = build(g)
    # same seed -> same initial networks as the reference built, including the unsynchronised target heads. (orthogonal_
    # runs a LAPACK QR on the host, so across different CPUs the draws agree to rounding, not bitwise: compare with a
    # tolerance, then load the fixture's exact weights for the stepping part.)
    for grp, mod in (("actor/", policy.actor), ("critic/", policy.critic), ("actor_tgt/", policy.target_actor), ("critic_tgt/", policy.target_critic)):
        got = params_of(mod)
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(got[k], ref, rtol=0, atol=3e-5, err_msg=grp + k)
        mod.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, grp).items()})
    np.testing.assert_allclose(policy.critic._head_w.cpu().numpy(), g["heads/w"], atol=3e-5)
    np.testing.assert_allclose(policy.target_critic._head_w.cpu().numpy(), g["heads_tgt/w"], atol=3e-5)
    for crit, pre in ((policy.critic, "heads/"), (policy.target_critic, "heads_tgt/")):
        crit._head_w.copy_(torch.as_tensor(g[pre + "w"]))
        crit._head_b.copy_(torch.as_tensor(g[pre + "b"]))
    assert list(policy.critic.state_dict().keys()) == list(sub(g, "critic/").keys())     # heads absent, as upstream
    s = buf.policy_buffers["policy_0"].sample_inds(g["inds"])
    w = g["per_weights"] if "per_weights" in g else None
    batch = tuple({"policy_0": a} for a in s) + (w, g["inds"] if w is not None else None)
    for st in range(len(g["critic_loss"])):
        torch.manual_seed(1000 + st)
        info, prio, _ = trainer.shared_train_policy_on_batch("policy_0", batch)
        assert info["update_actor"]
        policy.soft_target_updates()
        np.testing.assert_allclose(float(info["critic_loss"]), g["critic_loss"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["critic_grad_norm"]), g["critic_grad_norm"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["actor_loss"]), g["actor_loss"][st], rtol=5e-4, atol=2e-6)
        np.testing.assert_allclose(float(info["actor_grad_norm"]), g["actor_grad_norm"][st], rtol=5e-4)
        if w is not None:
            np.testing.assert_allclose(prio, g["priorities"][st], rtol=RTOL)
    for grp, mod in (("final_actor/", policy.actor), ("final_critic/", policy.critic), ("final_actor_tgt/", policy.target_actor),
                     ("final_critic_tgt/", policy.target_critic)):
        got = params_of(mod)
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(got[k], ref, rtol=0, atol=3e-5, err_msg=grp + k)
    assert np.array_equal(policy.critic._head_w.cpu().numpy(), g["final_heads/w"])          # frozen (A-4)
    assert np.array_equal(policy.target_critic._head_w.cpu().numpy(), g["final_heads_tgt/w"])


@pytest.mark.parametrize("name", ["maddpg_cent_small", "maddpg_cent_huber_per"])
def test_cent_train_policy_on_batch_matches_reference(name):
    """use_same_share_obs = False: every agent has its own centralized observation (MADDPG.cent_train_policy_on_batch,
    maddpg.py:251-419). Upstream the function crashes on the critic's list of heads (A-5); the fixtures are the reference
    run with the one-head reading documented in oracle/make_golden_cent.py. Buffer ([N, B, S] centralized observations out of the
    per-agent ring), critic on the N*B rows, priorities averaged over the agents, actor copies on their own agent's observation."""
    g = load_golden(name)
    dims, buf, policy, trainer = build(g, same_share=False)
    for grp, mod in (("actor/", policy.actor), ("critic/", policy.critic), ("actor_tgt/", policy.target_actor), ("critic_tgt/", policy.target_critic)):
        mod.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, grp).items()})
    for crit, pre in ((policy.critic, "heads/"), (policy.target_critic, "heads_tgt/")):
        crit._head_w.copy_(torch.as_tensor(g[pre + "w"]))
        crit._head_b.copy_(torch.as_tensor(g[pre + "b"]))
    s = buf.policy_buffers["policy_0"].sample_inds(g["inds"])
    for k, a in zip(T_KEYS, s):         # the per-agent centralized observations come back as the reference's buffer returns them
        if a is not None:
            assert tuple(a.shape) == g["batch/" + k].shape and np.array_equal(a.cpu().numpy(), g["batch/" + k]), k
    w = g["per_weights"] if "per_weights" in g else None
    batch = tuple({"policy_0": a} for a in s) + (w, g["inds"] if w is not None else None)
    for st in range(len(g["critic_loss"])):
        torch.manual_seed(1000 + st)
        info, prio, _ = trainer.train_policy_on_batch("policy_0", batch)        # dispatches on use_same_share_obs, as maddpg.py:83-88
        assert info["update_actor"]
        policy.soft_target_updates()
        np.testing.assert_allclose(float(info["critic_loss"]), g["critic_loss"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["critic_grad_norm"]), g["critic_grad_norm"][st], rtol=RTOL)
        np.testing.assert_allclose(float(info["actor_loss"]), g["actor_loss"][st], rtol=5e-4, atol=2e-6)
        np.testing.assert_allclose(float(info["actor_grad_norm"]), g["actor_grad_norm"][st], rtol=5e-4)
        if w is not None:
            np.testing.assert_allclose(prio, g["priorities"][st], rtol=RTOL)
    for grp, mod in (("final_actor/", policy.actor), ("final_critic/", policy.critic), ("final_actor_tgt/", policy.target_actor),
                     ("final_critic_tgt/", policy.target_critic)):
        got = params_of(mod)
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(got[k], ref, rtol=0, atol=3e-5, err_msg=grp + k)


def test_fixed_mode_trains_heads_and_delays_actor():
    """The documented non-reference mode: registered (trainable) Q heads and a real update counter."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims, policy_info_for
    from offpolicy_amd.algorithms.matd3.algorithm.MATD3Policy import MATD3Policy
    from offpolicy_amd.algorithms.matd3.matd3 import MATD3
    g = load_golden("matd3_small")
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    dims = EnvDims("fx", n, a, d, s, 1)
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    policy = MATD3Policy({"args": default_args(), "device": dev}, policy_info_for(dims)["policy_0"], frozen_q_head=False)
    trainer = MATD3(default_args(), n, {"policy_0": policy}, lambda x: "policy_0", device=dev, count_updates=True)
    assert "q_outs.weight" in policy.critic.state_dict()
    assert torch.equal(policy.critic._head_w, policy.target_critic._head_w)          # targets synchronised, heads included
    batch = tuple({"policy_0": g["batch/" + k]} for k in T_KEYS) + (None, None)
    h0 = policy.critic._head_w.clone()
    info0, _, _ = trainer.shared_train_policy_on_batch("policy_0", batch)
    info1, _, _ = trainer.shared_train_policy_on_batch("policy_0", batch)
    assert "actor_loss" in info0 and "actor_loss" not in info1                       # every 2nd update only
    assert not torch.equal(policy.critic._head_w, h0)
    assert np.isfinite(float(info1["critic_loss"]))


def test_graphed_step_matches_eager_step():
    """The captured HIP graph of one update (gather + critic + actor + soft updates) replays the same computation as the
    eager calls: first-replay critic loss / grad norm equal the eager ones on the same indices, Adam's device step counter
    advances, and repeated replays keep training."""
    import copy
    g = load_golden("maddpg_spread")
    res = {}
    for mode in ("eager", "graph"):
        dims, buf, policy, trainer = build(g)
        trainer.device_noise = True
        inds = np.asarray(g["inds"])
        if mode == "eager":
            s = buf.policy_buffers["policy_0"].sample_inds(inds)
            info, _, _ = trainer.shared_train_policy_on_batch("policy_0", tuple({"policy_0": a} for a in s) + (None, None))
            res[mode] = (float(info["critic_loss"]), float(info["critic_grad_norm"]))
        else:
            theta0 = {k: m._flat.clone() for k, m in (("a", policy.actor), ("c", policy.critic), ("ta", policy.target_actor), ("tc", policy.target_critic))}
            rng_before = np.random.get_state()[1].copy()
            step = trainer.make_graphed_step(buf, len(inds))
            # the capture warm-up trains two throw-away steps; make_graphed_step must put everything back
            for k, m in (("a", policy.actor), ("c", policy.critic), ("ta", policy.target_actor), ("tc", policy.target_critic)):
                assert torch.equal(m._flat, theta0[k]), k
            for opt in (policy.critic_optimizer, policy.actor_optimizer):
                assert not opt.exp_avg.any() and not opt.exp_avg_sq.any() and int(opt.step_dev[0].item()) == 0 and opt.step_count == 0
            assert np.array_equal(np.random.get_state()[1], rng_before)
            info = step(inds)
            res[mode] = (float(info["critic_loss"]), float(info["critic_grad_norm"]))
            assert int(policy.critic_optimizer.step_dev[0].item()) == 1 and policy.critic_optimizer.step_count == 1
            before = policy.actor._flat.clone()
            losses = [float(step(np.random.RandomState(s).choice(len(buf), len(inds)))["critic_loss"]) for s in range(5)]
            assert np.all(np.isfinite(losses)) and int(policy.actor_optimizer.step_dev[0].item()) == 6
            assert not torch.equal(before, policy.actor._flat)
    np.testing.assert_allclose(res["graph"], res["eager"], rtol=1e-5)
    np.testing.assert_allclose(res["eager"][0], g["critic_loss"][0], rtol=RTOL)


@pytest.mark.parametrize("td3", [False, True])
def test_graphed_prioritized_step_matches_eager_device_tree_step(td3):
    """Prioritized replay inside the captured graph: device trees, masses from torch.rand, `filled` and beta
    read from HBM (ope_per_tree_sample_dev). The replays draw, train and write priorities back exactly like the eager loop over
    the same device-tree calls started from the same generator state; building the graph leaves trees and generator untouched."""
    from offpolicy_amd.utils.mlp_buffer import PrioritizedMlpReplayBuffer
    from offpolicy_amd.utils.synth import policy_info_for
    res = {}
    for mode in ("eager", "graph"):
        dims, N, A, D, S, B, cap, policy, trainer, buf0 = config3_setup(td3, True, B=64, cap=512)
        pinfo = policy_info_for(dims)
        buf = PrioritizedMlpReplayBuffer(0.6, pinfo, {"policy_0": list(range(N))}, cap, True, True, False, device=policy.actor._flat.device,
                                         device_tree=True)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import ddpg_transitions, DDPG_KEYS
        data = ddpg_transitions(np.random.RandomState(5), cap - 100, dims)          # 412 of 512 slots filled
        buf.insert(cap - 100, *[{"policy_0": data[k]} for k in DDPG_KEYS])
        # uneven priorities to start from
        buf.update_priorities(np.arange(cap - 100), np.random.RandomState(6).uniform(0.1, 3.0, cap - 100).astype(np.float32), p_id="policy_0")
        trainer.device_noise = True
        torch.cuda.manual_seed(1234)
        betas = [0.4, 0.5, 0.6]
        drawn, prios = [], []
        if mode == "eager":
            beta_dev = torch.ones(1, dtype=torch.float64, device=policy.actor._flat.device)
            trainer.fuse_soft_update = True
            for opt in (policy.critic_optimizer, policy.actor_optimizer):      # device step counters, as the graphed step keeps them
                opt.step_dev = torch.tensor([opt.step_count, 0], dtype=torch.int32, device=policy.actor._flat.device)      # (they key the noise)
            for beta in betas:
                beta_dev.fill_(beta)
                batch = buf.sample_device(B, beta_dev, p_id="policy_0")
                info, prio, idx = trainer.shared_train_policy_on_batch("policy_0", batch)
                buf.update_priorities(idx, prio, p_id="policy_0")
                policy.soft_target_updates()
                drawn.append(idx.cpu().numpy()); prios.append(prio.cpu().numpy())
        else:
            trees0 = buf._dtrees["policy_0"].trees.clone()
            rng0 = torch.cuda.get_rng_state()
            theta0 = policy.critic._flat.clone()
            step = trainer.make_graphed_step(buf, B)
            assert torch.equal(buf._dtrees["policy_0"].trees, trees0) and torch.equal(torch.cuda.get_rng_state(), rng0)
            assert torch.equal(policy.critic._flat, theta0)
            for beta in betas:
                info = step(beta)
                drawn.append(info["indices"].cpu().numpy()); prios.append(info["priorities"].cpu().numpy())
        torch.cuda.synchronize()
        leaves = buf._dtrees["policy_0"].leaves()[0]
        res[mode] = (np.asarray(drawn), np.asarray(prios), policy.critic._flat.cpu().numpy(), policy.actor._flat.cpu().numpy(), leaves)
        assert np.all(res[mode][0] >= 0) and np.all(res[mode][0] < cap - 100)
        # the leaves of the last draw hold its priorities ** alpha (duplicates: the last one wins)
        last = {int(i): float(p) for i, p in zip(res[mode][0][-1], res[mode][1][-1])}
        for i, p in last.items():
            np.testing.assert_allclose(leaves[i], p ** buf.alpha, rtol=1e-6)
    e, g = res["eager"], res["graph"]
    assert np.array_equal(e[0], g[0]), "the graph replays must draw the indices the eager calls draw from the same generator state"
    np.testing.assert_allclose(g[1], e[1], rtol=1e-5)
    np.testing.assert_allclose(g[2], e[2], rtol=0, atol=2e-6)
    np.testing.assert_allclose(g[3], e[3], rtol=0, atol=2e-6)
    np.testing.assert_allclose(g[4], e[4], rtol=1e-5)


def _flat_grads(mod, gvec, n_tail):
    """Per-tensor normalised gradients out of a flat gradient vector; tail at n_tail = [loss_sum, count, ...]."""
    g = gvec.cpu().numpy()
    cnt = g[n_tail + 1]
    return {name: g[off:off + int(np.prod(shape))].reshape(shape) / cnt for name, (shape, off) in mod.spec().items()}


def config3_setup(td3, per, dims=None, B=256, cap=1024):
    """MADDPG / MATD3 trainer (default: MPE simple_spread dims, B = 256) with a filled buffer and perturbed networks."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import DIMS, policy_info_for
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy
    from offpolicy_amd.algorithms.matd3.algorithm.MATD3Policy import MATD3Policy
    from offpolicy_amd.algorithms.maddpg.maddpg import MADDPG
    from offpolicy_amd.algorithms.matd3.matd3 import MATD3
    dims = dims or DIMS["simple_spread"]
    N, A, D, S = dims.n_agents, dims.act_dim, dims.obs_dim, dims.state_dim
    args = default_args(use_per=per)
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    np.random.seed(1)
    pinfo = policy_info_for(dims)
    policy = (MATD3Policy if td3 else MADDPGPolicy)({"args": args, "device": dev}, pinfo["policy_0"])
    trainer = (MATD3 if td3 else MADDPG)(args, N, {"policy_0": policy}, lambda x: "policy_0", device=dev)
    rng = np.random.RandomState(21)
    f = np.float32
    dones_env = (rng.random_sample((cap, 1)) < 0.1).astype(f)
    avail = (rng.random_sample((cap, N, A)) < 0.8).astype(f)
    avail[..., 0] = 1.0
    navail = (rng.random_sample((cap, N, A)) < 0.8).astype(f)
    navail[..., 0] = 1.0
    tr = dict(obs=rng.standard_normal((cap, N, D)).astype(f), share_obs=rng.standard_normal((cap, S)).astype(f),
              acts=np.eye(A, dtype=f)[rng.randint(0, A, size=(cap, N))], rewards=np.repeat(rng.standard_normal((cap, 1, 1)).astype(f), N, 1),
              next_obs=rng.standard_normal((cap, N, D)).astype(f), next_share_obs=rng.standard_normal((cap, S)).astype(f),
              dones=np.repeat(dones_env[:, None], N, 1), dones_env=dones_env,
              valid_transition=(rng.random_sample((cap, N, 1)) < 0.9).astype(f), avail_acts=avail, next_avail_acts=navail)
    buf = MlpReplayBuffer(pinfo, {"policy_0": list(range(N))}, cap, True, True, False, device=dev)
    buf.insert(cap, *[{"policy_0": tr[k]} for k in T_KEYS])
    gen = torch.Generator(device="cpu").manual_seed(3)
    for mod in (policy.critic, policy.actor):      # away from the targets and the gain-0.01 output layers
        mod._flat.add_((0.05 * torch.randn(mod._flat.numel(), generator=gen)).to(dev))
    torch.cuda.synchronize()
    return dims, N, A, D, S, B, cap, policy, trainer, buf


@pytest.mark.parametrize("td3,per", [(False, False), (True, True)])
def test_config3_batch256_matches_oracle(td3, per):
    """BASELINE config 3 AT ITS OWN SIZE: MADDPG-MLP (and MATD3-MLP + prioritized replay) on MPE simple_spread
    (N=3, A=5, D=18, S=54), B=256 transitions: 768 actor rows / 256 and 768 critic rows per update, i.e. the row counts
    the benchmark runs at (the reference fixtures stop at B=32). Two updates vs oracle/maddpg_oracle.py on the
    reference's gumbel noise stream: losses, gradient norms, priorities, every gradient tensor of the first update and
    the parameters after both. (reference: maddpg.py:90-249)"""
    from oracle import maddpg_oracle as DO
    from oracle.qmix_oracle import HP
    dims, N, A, D, S, B, cap, policy, trainer, buf = config3_setup(td3, per)
    f = np.float32
    heads = lambda c: (c._head_w.cpu().numpy().reshape(-1, 64).copy(), c._head_b.cpu().numpy().reshape(-1).copy())
    orc = DO.MaddpgOracle(params_of(policy.actor), params_of(policy.critic), heads(policy.critic), params_of(policy.target_actor),
                          params_of(policy.target_critic), heads(policy.target_critic), N, HP(use_per=per), td3=td3)
    for st in range(2):
        inds = np.random.RandomState(30 + st).choice(cap, B)
        w = np.random.RandomState(40 + st).uniform(0.4, 1.0, size=B).astype(f) if per else None
        s = buf.policy_buffers["policy_0"].sample_inds(inds)
        np_batch = tuple(a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a) for a in s)
        torch.manual_seed(1000 + st)
        info, prio, _ = trainer.shared_train_policy_on_batch("policy_0", tuple({"policy_0": a} for a in s) + (w, inds if per else None))
        policy.soft_target_updates()
        torch.cuda.synchronize()
        torch.manual_seed(1000 + st)
        u_t = torch.FloatTensor(N * B, A).uniform_() if td3 else None
        u_a = torch.FloatTensor(N * B, A).uniform_()
        ref = orc.train_step(np_batch, u_t, u_a, weights=w)
        np.testing.assert_allclose(float(info["critic_loss"]), ref["critic_loss"], rtol=RTOL)
        np.testing.assert_allclose(float(info["critic_grad_norm"]), ref["critic_grad_norm"], rtol=RTOL)
        np.testing.assert_allclose(float(info["actor_loss"]), ref["actor_loss"], rtol=5e-4, atol=2e-6)
        np.testing.assert_allclose(float(info["actor_grad_norm"]), ref["actor_grad_norm"], rtol=5e-4)
        if per:
            np.testing.assert_allclose(np.asarray(prio), ref["priorities"], rtol=RTOL)
        if st == 0:
            gc, ga, _ = trainer._grads[B]
            for mod, gvec, rg in ((policy.critic, gc, ref["critic_grads"]), (policy.actor, ga, ref["actor_grads"])):
                got = _flat_grads(mod, gvec, mod.padded_numel)
                from golden_util import assert_grads_close
                assert_grads_close("ddpg_config3:td3=%d" % int(td3), got, rg, GRAD_TOL, skip_missing=True)
    for mod, refp in ((policy.critic, orc.critic), (policy.actor, orc.actor), (policy.target_critic, orc.critic_tgt),
                      (policy.target_actor, orc.actor_tgt)):
        for k, v in params_of(mod).items():
            np.testing.assert_allclose(v, refp[k].numpy(), rtol=0, atol=3e-5, err_msg=k)


def test_soft_and_hard_update_helpers_act_per_parameter():
    """utils.util.soft_update / hard_update (reference util.py:123-146) Polyak the registered parameters of the module
    they are given and nothing else: the frozen (unregistered, A-4) Q heads of a MADDPG critic keep their target values,
    and a policy object (anything with .parameters()) is accepted like upstream's soft_update(target_policy, policy, tau)."""
    from offpolicy_amd.utils.util import soft_update, hard_update
    g = load_golden("matd3_small")
    dims, buf, policy, trainer = build(g)
    policy.critic._flat.add_(0.5)
    policy.actor._flat.add_(0.25)
    tgt0, src = policy.target_critic._flat.clone(), policy.critic._flat.clone()
    soft_update(policy.target_critic, policy.critic, 0.1)
    n = policy.critic.head_offset
    want = tgt0.clone()
    for name, (shape, off) in policy.critic.spec().items():
        k = int(np.prod(shape))
        want[off:off + k] = tgt0[off:off + k] * 0.9 + src[off:off + k] * 0.1
    np.testing.assert_allclose(policy.target_critic._flat.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(policy.target_critic._flat[n:], tgt0[n:])            # frozen heads (and padding) untouched
    hard_update(policy.target_actor, policy.actor)
    for (k, a), (_, b) in zip(policy.target_actor.named_parameters(), policy.actor.named_parameters()):
        assert torch.equal(a, b), k


def _noise_worker(out_path):
    """One MATD3 update at config-3 size with the gumbel noise drawn inside the kernels; dumps the gradients (and, on the
    unfused path, the sampled actions and their logits)."""
    dims, N, A, D, S, B, cap, policy, trainer, buf = config3_setup(True, False)
    trainer.device_noise = True
    inds = np.random.RandomState(30).choice(cap, B)
    s = buf.policy_buffers["policy_0"].sample_inds(inds)
    torch.manual_seed(77)
    trainer.shared_train_policy_on_batch("policy_0", tuple({"policy_0": a} for a in s) + (None, None))
    torch.cuda.synchronize()
    gc, ga, _ = trainer._grads[B]
    out = dict(gc=gc.cpu().numpy(), ga=ga.cpu().numpy())
    try:
        out["act_out"] = trainer.workspace_view(B, "act_out").cpu().numpy().reshape(N * B, A)
        out["logits"] = trainer.workspace_view(B, "logits").cpu().numpy().reshape(N * B, A)
        out["avail"] = s[9].cpu().numpy().reshape(N * B, A)
    except KeyError:
        pass
    np.savez(out_path, **out)


def test_device_noise_same_stream_on_both_paths_and_right_distribution(tmp_path):
    """`device_noise`: the kernels draw the gumbel noise themselves (Philox4x32-10 keyed by seed, step count, row, column).
    (1) The fused small-network path and the general path (OPE_DDPG_FUSED=0) use the SAME stream: their critic and actor
    gradients agree; this also runs the general path at config-3 size, which the default dispatch never takes.
    (2) The hard gumbel-softmax actions the general path leaves in its workspace are distributed as softmax(masked logits)
    (util.py:193-207): per action, the sampled count is within 4.5 sigma of its expectation over the 768 rows."""
    import subprocess, sys
    outs = {}
    for fused in ("1", "0"):
        path = str(tmp_path / ("g%s.npz" % fused))
        env = dict(os.environ, OPE_DDPG_FUSED=fused)
        code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_ddpg as t; t._noise_worker(%r)" % (
            os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[fused] = np.load(path)
    for k in ("gc", "ga"):
        a, b = outs["1"][k], outs["0"][k]
        assert np.abs(b).max() > 0
        from golden_util import assert_grads_close
        assert_grads_close("ddpg_fused_vs_separate", {k: a}, {k: b}, GRAD_TOL)
    act, lg, av = outs["0"]["act_out"], outs["0"]["logits"], outs["0"]["avail"]
    onehot = (act > 0.5).astype(np.float64)          # straight-through value: 1 at the sampled action (+- float noise)
    assert np.all(onehot.sum(1) == 1) and np.all(onehot * (1 - av) == 0)
    z = np.where(av > 0, lg.astype(np.float64), -1e10)
    p = np.exp(z - z.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    mean, var = p.sum(0), (p * (1 - p)).sum(0)
    assert np.all(np.abs(onehot.sum(0) - mean) < 4.5 * np.sqrt(var) + 1e-9), (onehot.sum(0), mean)


def test_general_path_matches_reference_fixtures():
    """The small-network dispatch sends every fixture through the fused kernels; the general (tile/MFMA trunk) path, which
    larger observation / joint-action widths take, is run here on the same reference fixtures with OPE_DDPG_FUSED=0."""
    import subprocess, sys
    if os.environ.get("OPE_DDPG_FUSED") == "0":
        pytest.skip("already on the general path")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "construction_and_train_steps or config3_batch256", "-p", "no:cacheprovider"],
                       env=dict(os.environ, OPE_DDPG_FUSED="0"), capture_output=True, text=True, timeout=1200,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


def test_device_sampled_graph_steps_are_the_same_alone_or_batched():
    """Graph replay with everything drawn on the device (batch indices inside the gather, gumbel noise inside the update
    kernels, Adam step counts): three replays of a one-step graph and ONE replay of a three-step graph leave bit-identical
    networks, targets and Adam state (every kernel is deterministic and the Philox streams are keyed by the device step
    count), the drawn indices lie inside the filled part of the buffer, and the step counts advance by three."""
    finals = []
    for spr in (1, 3):
        dims, N, A, D, S, B, cap, policy, trainer, buf = config3_setup(True, False)
        torch.manual_seed(5)
        step = trainer.make_graphed_step(buf, B, device_sampling=True, steps_per_replay=spr)
        for _ in range(3 // spr):
            info = step()
        torch.cuda.synchronize()
        iv = info["indices"].cpu().numpy()
        assert iv.min() >= 0 and iv.max() < len(buf) and len(np.unique(iv)) > B // 2
        assert int(policy.critic_optimizer.step_dev[0].item()) == 3 == policy.critic_optimizer.step_count
        assert np.isfinite(float(info["critic_loss"])) and np.isfinite(float(info["actor_loss"]))
        finals.append([m._flat.clone() for m in (policy.actor, policy.critic, policy.target_actor, policy.target_critic)] +
                      [policy.critic_optimizer.exp_avg.clone(), policy.actor_optimizer.exp_avg_sq.clone(), info["indices"].clone()])
    for a, b in zip(*finals):
        assert torch.equal(a, b)


def _shape_worker(out_path, n, a, d, s_dim, B, td3):
    """One update with in-kernel noise on arbitrary dimensions; dumps both gradients, the losses and the priorities."""
    from offpolicy_amd.utils.synth import EnvDims
    dims, N, A, D, S, B, cap, policy, trainer, buf = config3_setup(bool(td3), True, dims=EnvDims("t", n, a, d, s_dim, 1), B=B, cap=max(1024, B))
    trainer.device_noise = True
    inds = np.random.RandomState(30).choice(cap, B)
    w = np.random.RandomState(40).uniform(0.4, 1.0, size=B).astype(np.float32)
    sm = buf.policy_buffers["policy_0"].sample_inds(inds)
    torch.manual_seed(77)
    info, prio, _ = trainer.shared_train_policy_on_batch("policy_0", tuple({"policy_0": x} for x in sm) + (w, inds))
    torch.cuda.synchronize()
    gc, ga, _ = trainer._grads[B]
    np.savez(out_path, gc=gc.cpu().numpy(), ga=ga.cpu().numpy(), prio=np.asarray(prio), closs=float(info["critic_loss"]),
             aloss=float(info["actor_loss"]), theta_a=policy.actor._flat.cpu().numpy(), theta_c=policy.critic._flat.cpu().numpy())


@pytest.mark.parametrize("n,a,d,s_dim,B,td3", [
    (5, 7, 40, 90, 100, 1),      # actor bucket 4 chunks, critic 8 (Din = 125), B not a multiple of the 16-row tile, 5 agents (two input batches)
    (6, 3, 100, 60, 48, 0),      # actor bucket 8 chunks (D = 100), critic bucket 5 (Din = 78), 6 agents
    (1, 16, 20, 12, 33, 1),      # one agent, 16 actions (a full head tile), ragged last tile
    (2, 4, 10, 16, 9000, 0),     # 1 125 actor tiles > 1 024 workgroups: the grid-stride tile loop and slab accumulation
])
def test_fused_tile_path_matches_general_path_on_other_shapes(tmp_path, n, a, d, s_dim, B, td3):
    """The MFMA tile kernels (default for small networks) against the general launch sequence (OPE_DDPG_FUSED=0) of the same
    library on shapes the reference fixtures do not cover: every template bucket of the input width, agent counts beyond one
    staging batch, batches that do not fill the last 16-row tile, a 16-action head, and more tiles than workgroups. Same
    in-kernel noise stream, prioritized weights: gradients, losses, priorities and the parameters after the update agree."""
    import subprocess, sys
    outs = {}
    for fused in ("1", "0"):
        path = str(tmp_path / ("s%s.npz" % fused))
        code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_ddpg as t; t._shape_worker(%r, %d, %d, %d, %d, %d, %d)" % (
            os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, n, a, d, s_dim, B, td3)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OPE_DDPG_FUSED=fused), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[fused] = np.load(path)
    f, g = outs["1"], outs["0"]
    np.testing.assert_allclose(f["closs"], g["closs"], rtol=2e-4)
    np.testing.assert_allclose(f["aloss"], g["aloss"], rtol=5e-4, atol=2e-6)
    np.testing.assert_allclose(f["prio"], g["prio"], rtol=2e-4, atol=1e-6)
    for k in ("gc", "ga"):
        assert np.abs(g[k]).max() > 0
        from golden_util import assert_grads_close
        assert_grads_close("ddpg_graph_vs_eager", {k: f[k]}, {k: g[k]}, GRAD_TOL)
    for k in ("theta_a", "theta_c"):
        np.testing.assert_allclose(f[k], g[k], rtol=0, atol=5e-5, err_msg=k)


@pytest.mark.parametrize("name", ["maddpg_spread", "matd3_spread", "maddpg_small_huber_per", "maddpg_small_wd"])
def test_optimiser_step_inside_the_update_launch_matches_the_separate_launches(name):
    """ope_ddpg_critic_update / ope_ddpg_actor_update (round 4): slab reduction, clip_grad_norm_, Adam and the
    target's Polyak step in the tail of the tile launch, behind two grid barriers. Against the separate launches (tile kernel, slab
    reduction, ope_adam_step -- what the reference fixtures above were matched with): the gradient vectors are bit-identical (same
    summation order), losses and norms equal to float rounding of the norm's partial sums, parameters / targets / Adam moments within
    1e-6 after three steps, the grid barrier never timed out, and the fixture's own numbers hold. maddpg.py:100-157, 192-249.
    (Opt-in: measured slower than the separate launches on MI355X, see csrc/ope_ddpg_tile.hip; kept correct.)"""
    from offpolicy_amd import _lib
    g = load_golden(name)
    runs = []
    for in_launch in (False, True):
        dims, buf, policy, trainer = build(g)
        for grp, mod in (("actor/", policy.actor), ("critic/", policy.critic), ("actor_tgt/", policy.target_actor), ("critic_tgt/", policy.target_critic)):
            mod.load_state_dict({k: torch.as_tensor(v) for k, v in sub(g, grp).items()})
        for crit, pre in ((policy.critic, "heads/"), (policy.target_critic, "heads_tgt/")):
            crit._head_w.copy_(torch.as_tensor(g[pre + "w"]))
            crit._head_b.copy_(torch.as_tensor(g[pre + "b"]))
        trainer.update_in_launch = in_launch
        s = buf.policy_buffers["policy_0"].sample_inds(g["inds"])
        w = g["per_weights"] if "per_weights" in g else None
        batch = tuple({"policy_0": a} for a in s) + (w, g["inds"] if w is not None else None)
        B = len(g["inds"])
        assert trainer._update_in_launch(policy.ddpg_cfg(B)) == in_launch
        infos, grads = [], []
        for st in range(len(g["critic_loss"])):
            torch.manual_seed(1000 + st)
            info, prio, _ = trainer.shared_train_policy_on_batch("policy_0", batch)
            policy.soft_target_updates()
            infos.append({k: float(v) for k, v in info.items() if k != "update_actor"})
            gc, ga, _ = trainer._grads[B]
            grads.append((gc.clone(), ga.clone()))
            np.testing.assert_allclose(infos[-1]["critic_loss"], g["critic_loss"][st], rtol=RTOL)
            np.testing.assert_allclose(infos[-1]["actor_grad_norm"], g["actor_grad_norm"][st], rtol=5e-4)
        if in_launch:
            assert int(trainer.workspace_view(B, "opt_sync").view(torch.int32)[0].item()) == 0, "a grid barrier of the update launch timed out"
        runs.append((infos, grads, [x._flat.clone() for x in (policy.actor, policy.critic, policy.target_actor, policy.target_critic)],
                     [o.exp_avg.clone() for o in (policy.actor_optimizer, policy.critic_optimizer)]))
    (i0, g0, p0, m0), (i1, g1, p1, m1) = runs
    assert torch.equal(g0[0][0], g1[0][0]) and torch.equal(g0[0][1][:4], g1[0][1][:4])      # first step: identical inputs -> identical critic gradient
    for a, b in zip(i0, i1):
        for k in a:
            np.testing.assert_allclose(a[k], b[k], rtol=2e-6, atol=1e-9, err_msg=k)
    for a, b in zip(p0 + m0, p1 + m1):
        assert float((a - b).abs().max()) <= 1e-6
