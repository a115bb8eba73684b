"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY. Run in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Each fixture freezes: initial weights of the reference's modules, the synthetic episodes (or the seed +
a digest when they are large), the indices sampled, what `RecPolicyBuffer.sample_inds` returned, and for
each of K train steps the reference's loss / grad_norm / Q_tot / priorities plus the post-step weights of
the live and target networks. The reference is unmodified except for the documented oracle patch for VDN
(`VDNMixer.forward` must keep [T,B,1]).
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference, reference_args  # noqa: E402

load_reference()
from gym.spaces import Discrete  # noqa: E402  (the stub; the reference's isinstance checks need this class)
from offpolicy.utils.rec_buffer import RecReplayBuffer, PrioritizedRecReplayBuffer  # noqa: E402
from offpolicy.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy  # noqa: E402
from offpolicy.algorithms.qmix.qmix import QMix  # noqa: E402
import offpolicy.algorithms.vdn.algorithm.vdn_mixer as vdn_mixer_mod  # noqa: E402

from offpolicy_amd.utils.synth import DIMS, EnvDims, synth_episodes, as_policy_dicts  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def digest(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def patch_vdn():
    """Oracle patch A-2: keep the [T, B, 1] shape."""
    def forward(self, agent_q_inps, states):
        if type(agent_q_inps) == np.ndarray:
            agent_q_inps = torch.FloatTensor(agent_q_inps)
        return agent_q_inps.sum(dim=-1, keepdim=True).unsqueeze(-1)  # QMix squeezes the last dim (qmix.py:155)
    vdn_mixer_mod.VDNMixer.forward = forward


def build(dims, argv=(), vdn=False, multi_discrete=None, **over):
    args = reference_args(argv, **over)
    torch.manual_seed(1)
    np.random.seed(1)
    act_space = Discrete(dims.act_dim)
    if multi_discrete is not None:      # sub-action sizes (their sum = dims.act_dim): one q head each, one mixer input per (agent, sub-action)
        from offpolicy.utils.util import MultiDiscrete
        assert sum(multi_discrete) == dims.act_dim
        act_space = MultiDiscrete([[0, k - 1] for k in multi_discrete])
    pinfo = {"policy_0": {"cent_obs_dim": dims.state_dim, "cent_act_dim": dims.act_dim * dims.n_agents,
                          "obs_space": [dims.obs_dim], "share_obs_space": [dims.state_dim],
                          "act_space": act_space}}
    device = torch.device("cpu")
    policy = QMixPolicy({"args": args, "device": device}, pinfo["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=device,
                   episode_length=dims.episode_length, vdn=vdn)
    return args, pinfo, policy, trainer


def named(module, prefix):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.named_parameters()}


def run_case(name, dims, n_episodes, inds, steps=3, argv=(), vdn=False, avail="ones", runner_padding=False,
             per_weights=None, store_inputs=True, cap=None, pre_insert=0, per_agent_share=False, hard_update_after=(), multi_discrete=None, **over):
    """`hard_update_after`: with `--use_soft_update` given (which turns soft updates OFF, config.py:125) the runner copies the
    live networks into the targets every `hard_update_interval_episode` episodes instead (base_runner.py:279-284); here: after
    the listed (0-based) train steps."""
    if per_agent_share:      # every agent has its own centralized observation; QMix's mixer reads agent 0's (qmix.py:86-90)
        over = dict(over, use_same_share_obs=False)
    args, pinfo, policy, trainer = build(dims, argv, vdn=vdn, multi_discrete=multi_discrete, **over)
    cap = cap or n_episodes
    agents = {"policy_0": list(range(dims.n_agents))}
    # (MultiDiscrete spaces come without availability masks: upstream's avail_choose on the list of q heads fails)
    buf = RecReplayBuffer(pinfo, agents, cap, dims.episode_length, not per_agent_share, multi_discrete is None, False)
    rng = np.random.RandomState(0)
    out = {}
    if pre_insert:
        # exercise ring wrap-around: insert a first block that will be partly overwritten
        ep0 = synth_episodes(rng, pre_insert, dims, avail=avail, runner_padding=runner_padding)
        d0 = as_policy_dicts(ep0)
        rng0 = buf.insert(pre_insert, d0["obs"], d0["share_obs"], d0["acts"], d0["rewards"], d0["dones"],
                          d0["dones_env"], d0["avail_acts"])
        out["pre_idx_range"] = np.asarray(rng0)
        if store_inputs:
            for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts"):
                out["pre_ep/" + k] = ep0[k]
    ep = synth_episodes(rng, n_episodes, dims, avail=avail, runner_padding=runner_padding)
    if multi_discrete is not None:      # stored actions: one one-hot block per sub-action
        ep["acts"] = np.concatenate([np.eye(k, dtype=np.float32)[rng.randint(0, k, size=ep["acts"].shape[:3])] for k in multi_discrete], axis=-1)
        out["multi_discrete"] = np.asarray(multi_discrete, dtype=np.int64)
    if per_agent_share:      # make the agents' copies differ, so that using any other than agent 0's shows
        ep["share_obs"] = ep["share_obs"] + 0.25 * np.arange(dims.n_agents, dtype=np.float32)[None, None, :, None]
    d = as_policy_dicts(ep)
    idx_range = buf.insert(n_episodes, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"],
                           d["dones_env"], d["avail_acts"])
    out["idx_range"] = np.asarray(idx_range)
    out["lengths"] = ep["lengths"]
    out["filled_i"] = np.int64(buf.policy_buffers["policy_0"].filled_i)
    out["current_i"] = np.int64(buf.policy_buffers["policy_0"].current_i)
    keys = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")
    if store_inputs:
        for k in keys:
            out["ep/" + k] = ep[k]
    out["ep_digest"] = np.array(digest([ep[k] for k in keys]))
    out["ep_avail"], out["ep_runner_padding"] = np.array(avail), np.int64(bool(runner_padding))     # (what regenerates them when not stored)
    inds = np.asarray(inds, dtype=np.int64)
    out["inds"] = inds
    sampled = buf.policy_buffers["policy_0"].sample_inds(inds)
    if store_inputs:
        for k, a in zip(("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts"), sampled):
            if a is not None:
                out["batch/" + k] = np.ascontiguousarray(a)
    out["batch_digest"] = np.array(digest([a for a in sampled if a is not None]))
    out.update(named(policy.q_network, "agent/"))
    if not vdn:
        out.update(named(trainer.mixer, "mixer/"))
    out["dims"] = np.array([dims.n_agents, dims.act_dim, dims.obs_dim, dims.state_dim, dims.episode_length], dtype=np.int64)
    out["hp_gamma"], out["hp_lr"], out["hp_eps"] = np.float64(args.gamma), np.float64(args.lr), np.float64(args.opti_eps)
    out["hp_huber"], out["hp_delta"] = np.int64(args.use_huber_loss), np.float64(args.huber_delta)
    out["hp_per"], out["hp_nu"], out["hp_per_eps"] = np.int64(args.use_per), np.float64(args.per_nu), np.float64(args.per_eps)
    out["hp_tau"], out["hp_maxnorm"], out["hp_double_q"] = np.float64(args.tau), np.float64(args.max_grad_norm), np.int64(args.use_double_q)
    out["vdn"] = np.int64(vdn)
    out["hp_prev_act_inp"] = np.int64(bool(getattr(args, "prev_act_inp", False)))
    out["hp_same_share"] = np.int64(not per_agent_share)
    out["hp_soft_update"], out["hp_gain"] = np.int64(bool(args.use_soft_update)), np.float64(args.gain)
    out["hp_hidden_size"], out["hp_layer_N"], out["hp_hypernet_layers"] = np.int64(args.hidden_size), np.int64(args.layer_N), np.int64(args.hypernet_layers)
    out["hp_feature_norm"] = np.int64(bool(args.use_feature_normalization))
    out["hp_use_relu"] = np.int64(bool(args.use_ReLU))
    out["hard_update_after"] = np.asarray(list(hard_update_after), dtype=np.int64)
    losses, gnorms, qtots, prios = [], [], [], []
    for s in range(steps):
        batch = tuple({"policy_0": a} for a in sampled) + (per_weights, inds if per_weights is not None else None)
        info, new_prio, _ = trainer.train_policy_on_batch(batch)
        if s == 0:
            # param.grad after the in-place clip (qmix.py:192); fc_h stays None
            for k, v in policy.q_network.named_parameters():
                if v.grad is not None:
                    out["grad0/agent/" + k] = v.grad.detach().numpy().copy()
            if not vdn:
                for k, v in trainer.mixer.named_parameters():
                    out["grad0/mixer/" + k] = v.grad.detach().numpy().copy()
        if args.use_soft_update:
            trainer.soft_target_updates()
        elif s in hard_update_after:
            trainer.hard_target_updates()
        losses.append(float(info["loss"])), gnorms.append(float(info["grad_norm"])), qtots.append(float(info["Q_tot"]))
        if new_prio is not None:
            prios.append(np.asarray(new_prio, dtype=np.float64))
    out["loss"], out["grad_norm"], out["Q_tot"] = np.array(losses), np.array(gnorms), np.array(qtots)
    if prios:
        out["priorities"] = np.stack(prios)
        out["per_weights"] = np.asarray(per_weights)
    out.update(named(policy.q_network, "final_agent/"))
    out.update(named(trainer.target_policies["policy_0"].q_network, "final_agent_tgt/"))
    if not vdn:
        out.update(named(trainer.mixer, "final_mixer/"))
        out.update(named(trainer.target_mixer, "final_mixer_tgt/"))
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s loss=%s grad_norm=%s Q_tot=%s  -> %s (%.0f KB)" % (
        name, np.round(out["loss"], 7), np.round(out["grad_norm"], 6), np.round(out["Q_tot"], 7), path,
        os.path.getsize(path) / 1024.0))
    return out


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    tiny = DIMS["tiny"]
    if os.environ.get("OPE_GOLDEN_ONLY") == "multidiscrete":      # round 4: MultiDiscrete action spaces (oracle only; the engine refuses them)
        run_case("qmix_md_tiny", EnvDims("md_tiny", 2, 5, 6, 7, 4), n_episodes=5, inds=[0, 3, 1, 4], multi_discrete=[2, 3])
        run_case("qmix_md_odd_huber_per", EnvDims("md_odd", 3, 9, 7, 9, 6), n_episodes=6, inds=[1, 2, 5, 4], multi_discrete=[4, 2, 3],
                 argv=["--use_huber_loss", "--huber_delta", "0.5", "--use_per"], per_weights=np.array([0.3, 1.0, 0.6, 0.9]), steps=4)
        run_case("qmix_md_nodouble", EnvDims("md_tiny", 2, 5, 6, 7, 4), n_episodes=5, inds=[0, 3, 1, 4], multi_discrete=[2, 3], argv=["--use_double_q"])
        # (VDN: upstream's VDNMixer.forward with `multidiscrete_list` fails on the RNN trainer's [T, B, n] input, as A-2 does without -- no fixture)
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "gall":
        # round 3: the configuration the reference's ONLY QMIX-SMAC launch script runs (scripts/train_smac_qmix.sh:14-17):
        # --use_global_all_local_state (the centralized state carries every agent's observation too: S = s + N * D,
        # StarCraft2_Env.py:1314-1315), --gain 1 (head initialised with gain 1 instead of 0.01, act.py:10-12) and --use_soft_update
        # (a store_false flag: hard target copies every hard_update_interval_episode episodes instead of Polyak steps)
        gall = ["--gain", "1", "--use_soft_update"]
        t = tiny
        run_case("qmix_gall_tiny", EnvDims("tiny_gall", t.n_agents, t.act_dim, t.obs_dim, t.state_dim + t.n_agents * t.obs_dim, t.episode_length),
                 n_episodes=5, inds=[3, 0, 4, 3], cap=6, pre_insert=3, avail="bernoulli", steps=4, argv=gall, hard_update_after=(1,))
        m = DIMS["3m"]
        run_case("qmix_gall_3m", EnvDims("3m_gall", m.n_agents, m.act_dim, m.obs_dim, m.state_dim + m.n_agents * m.obs_dim, 12),
                 n_episodes=7, inds=[6, 1, 2, 2, 0, 5], avail="bernoulli", runner_padding=True, steps=4, argv=gall, hard_update_after=(1,))
        # an odd wide state (nothing a multiple of 4 or 2): the unaligned paths of the wide-state kernels
        run_case("qmix_gall_odd", EnvDims("odd_gall", 3, 7, 18, 29 + 3 * 18, 5), n_episodes=6, inds=[5, 1, 1, 2, 0], avail="bernoulli", steps=3,
                 argv=gall, hard_update_after=(0,))
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "shapes":
        # round 3: network shapes other than the reference defaults: one-layer hyper-networks (q_mixer.py:39-44),
        # two hidden blocks after fc1 (mlp.py:14-28), hidden_size 128. The ORACLE is pinned on them here; the HIP engine still
        # refuses them (config.py:require_reference_architecture) -- these fixtures are what its generic path will be tested on.
        run_case("qmix_shape_hyper1", tiny, n_episodes=5, inds=[3, 0, 4, 3], avail="bernoulli", argv=["--hypernet_layers", "1"])
        run_case("qmix_shape_layer2", tiny, n_episodes=5, inds=[1, 0, 4, 2], avail="bernoulli", argv=["--layer_N", "2"])
        run_case("qmix_shape_h128", tiny, n_episodes=5, inds=[2, 2, 0, 4], avail="bernoulli", argv=["--hidden_size", "128"])
        odd = EnvDims("odd", 3, 7, 18, 54, 5)
        run_case("qmix_shape_all_odd", odd, n_episodes=6, inds=[5, 1, 1, 2, 0], avail="bernoulli", runner_padding=True,
                 argv=["--hypernet_layers", "1", "--layer_N", "2", "--hidden_size", "128"])
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "variants":
        # round 4: fixtures that REACH the kernel variants bench.py runs. trunk_fwd4 only takes input widths that are
        # multiples of 4 with ceil(D / 16) in {4, 8, 12, 16}: D = 252 (KCM 16, with the 12-float tail chunk -- the 3s5z width), 188 (KCM 12),
        # 124 (KCM 8); 56 data rows = three full 16-row tiles + a partial one. mixer_fwd3 at the 3s5z state width with all eight agent waves
        # (S = 216, N = 8: the <14, FULL> instantiation) and at a narrower one (S = 100: the guarded K loop); the wide-state stream-K GEMM at
        # the real S = 2 232 (70 K stages of 32, a 24-float tail) and, with T * B = 156 rows, two 128-row blocks whose K ranges are cut
        # across workgroups (280 units on 256 workgroups).
        run_case("qmix_var_d252", EnvDims("var_d252", 2, 5, 252, 20, 6), n_episodes=5, inds=[4, 0, 2, 2], avail="bernoulli")
        run_case("qmix_var_d188", EnvDims("var_d188", 2, 5, 188, 20, 6), n_episodes=5, inds=[1, 3, 0, 4], avail="bernoulli", runner_padding=True)
        run_case("qmix_var_d124", EnvDims("var_d124", 3, 6, 124, 24, 6), n_episodes=5, inds=[2, 2, 1, 0], avail="bernoulli")
        run_case("qmix_var_mix216", EnvDims("var_mix216", 8, 6, 16, 216, 5), n_episodes=5, inds=[3, 1, 4, 0], avail="bernoulli")
        run_case("qmix_var_mix100", EnvDims("var_mix100", 5, 6, 16, 100, 5), n_episodes=5, inds=[0, 1, 4, 4, 2], avail="bernoulli", runner_padding=True)
        # more agents than the fused chain kernel has waves (two agents per wave) and more actions than one 16-action head tile
        run_case("qmix_var_n10", EnvDims("var_n10", 10, 18, 16, 40, 5), n_episodes=5, inds=[2, 0, 4, 1, 1], avail="bernoulli")
        run_case("qmix_var_a20", EnvDims("var_a20", 3, 20, 16, 24, 5), n_episodes=5, inds=[3, 3, 0, 1], avail="bernoulli", argv=["--use_double_q"])
        # one-layer hyper-networks (--hypernet_layers 1) beyond the tiny shape fixture: 8 agents at the 3s5z state width (2 N + 8 = 24 first-layer
        # tiles: three full column groups and a partial one) and an odd state width (scalar loads) with PER weights + Huber
        run_case("qmix_var_hyper1_mix", EnvDims("var_hyper1_mix", 8, 6, 16, 216, 5), n_episodes=5, inds=[3, 1, 4, 0], avail="bernoulli",
                 argv=["--hypernet_layers", "1"])
        run_case("qmix_var_hyper1_odd", EnvDims("var_hyper1_odd", 3, 7, 18, 29, 5), n_episodes=6, inds=[5, 1, 1, 2, 0], avail="bernoulli", runner_padding=True,
                 per_weights=np.array([1.0, 0.5, 0.25, 0.8, 0.9]), argv=["--hypernet_layers", "1", "--use_huber_loss", "--huber_delta", "1.0", "--use_per"])
        run_case("qmix_var_s2232", EnvDims("var_s2232", 2, 5, 12, 2232, 12), n_episodes=13, inds=list(range(13)), avail="bernoulli", steps=2,
                 store_inputs=False)
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "layer2":
        # round 4: --layer_N 2 beyond the tiny shape fixture: the 3s5z input width (the vectorised trunk kernels end at the first
        # block's output instead of the GRU projection), VDN, together with one-layer hyper-networks, and with the previous action as an input +
        # Huber + PER weights on odd sizes
        run_case("qmix_var_layer2_d252", EnvDims("var_layer2_d252", 2, 5, 252, 20, 6), n_episodes=5, inds=[4, 0, 2, 2], avail="bernoulli",
                 argv=["--layer_N", "2"])
        run_case("qmix_var_layer2_hyper1", EnvDims("var_layer2_hyper1", 5, 6, 16, 100, 5), n_episodes=5, inds=[0, 1, 4, 4, 2], avail="bernoulli",
                 runner_padding=True, argv=["--layer_N", "2", "--hypernet_layers", "1"])
        run_case("qmix_var_layer2_odd", EnvDims("var_layer2_odd", 3, 7, 18, 29, 5), n_episodes=6, inds=[5, 1, 1, 2, 0], avail="bernoulli",
                 runner_padding=True, per_weights=np.array([1.0, 0.5, 0.25, 0.8, 0.9]),
                 argv=["--layer_N", "2", "--prev_act_inp", "--use_huber_loss", "--huber_delta", "1.0", "--use_per"])
        patch_vdn()
        run_case("vdn_var_layer2", tiny, n_episodes=4, inds=[2, 1, 0, 3], avail="bernoulli", vdn=True, argv=["--layer_N", "2"])
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "d370":
        # round 4: the MMM2 observation width (D = 370: 24 chunks, rows 8-byte aligned only) on a small batch, for the LDS-resident trunk
        # kernel's 24-chunk / 8-byte-load instantiation (trunk_fwd4<24>, csrc/ope_trunk4.hip); 56 rows = three full tiles + a partial one
        run_case("qmix_var_d370", EnvDims("var_d370", 2, 5, 370, 20, 6), n_episodes=5, inds=[4, 0, 2, 2], avail="bernoulli")
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "nofn":
        # round 4 (tail): --use_feature_normalization (a store_false flag: no LayerNorm on the network input, mlp.py:60-62):
        # the tiny shape, the 3s5z width (the LDS-resident trunk kernels), odd sizes + previous action + Huber + PER, VDN, and together with
        # a second hidden block and one-layer hyper-networks
        run_case("qmix_shape_nofn", tiny, n_episodes=5, inds=[3, 1, 4, 0], avail="bernoulli", argv=["--use_feature_normalization"])
        run_case("qmix_var_nofn_d252", EnvDims("var_nofn_d252", 2, 5, 252, 20, 6), n_episodes=5, inds=[4, 0, 2, 2], avail="bernoulli",
                 argv=["--use_feature_normalization"])
        run_case("qmix_var_nofn_odd", EnvDims("var_nofn_odd", 3, 7, 18, 29, 5), n_episodes=6, inds=[5, 1, 1, 2, 0], avail="bernoulli",
                 runner_padding=True, per_weights=np.array([1.0, 0.5, 0.25, 0.8, 0.9]),
                 argv=["--use_feature_normalization", "--prev_act_inp", "--use_huber_loss", "--huber_delta", "1.0", "--use_per"])
        run_case("qmix_var_nofn_layer2_hyper1", EnvDims("var_nofn_l2h1", 5, 6, 16, 100, 5), n_episodes=5, inds=[0, 1, 4, 4, 2], avail="bernoulli",
                 runner_padding=True, argv=["--use_feature_normalization", "--layer_N", "2", "--hypernet_layers", "1"])
        patch_vdn()
        run_case("vdn_var_nofn", tiny, n_episodes=4, inds=[2, 1, 0, 3], avail="bernoulli", vdn=True, argv=["--use_feature_normalization"])
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "tanh":
        # round 4: --use_ReLU (a store_false flag: tanh behind fc1 / fc2 of the agent network, mlp.py:9-12; orthogonal init with the tanh gain):
        # the tiny shape, the 3s5z width, odd sizes + previous action + Huber + PER (with no input LayerNorm either), VDN
        run_case("qmix_shape_tanh", tiny, n_episodes=5, inds=[2, 0, 4, 1], avail="bernoulli", argv=["--use_ReLU"])
        run_case("qmix_var_tanh_d252", EnvDims("var_tanh_d252", 2, 5, 252, 20, 6), n_episodes=5, inds=[4, 0, 2, 2], avail="bernoulli", argv=["--use_ReLU"])
        run_case("qmix_var_tanh_odd", EnvDims("var_tanh_odd", 3, 7, 18, 29, 5), n_episodes=6, inds=[5, 1, 1, 2, 0], avail="bernoulli",
                 runner_padding=True, per_weights=np.array([1.0, 0.5, 0.25, 0.8, 0.9]),
                 argv=["--use_ReLU", "--use_feature_normalization", "--prev_act_inp", "--use_huber_loss", "--huber_delta", "1.0", "--use_per"])
        patch_vdn()
        run_case("vdn_var_tanh", tiny, n_episodes=4, inds=[2, 1, 0, 3], avail="bernoulli", vdn=True, argv=["--use_ReLU"])
        return
    if os.environ.get("OPE_GOLDEN_ONLY") == "pershare":      # add the round-2 fixture without rewriting the committed ones
        run_case("qmix_tiny_pershare", tiny, n_episodes=5, inds=[4, 1, 1, 0, 2], cap=6, pre_insert=3, avail="bernoulli", per_agent_share=True)
        return
    # 1. tiny dims, masked availability, MSE, ring wrap-around on insert, repeated index in the sample
    run_case("qmix_tiny", tiny, n_episodes=5, inds=[3, 0, 5, 3], cap=6, pre_insert=4, avail="bernoulli")
    # 2. tiny dims, Huber + PER weights + runner-style padding after the episode end
    run_case("qmix_tiny_huber_per", tiny, n_episodes=6, inds=[1, 4, 2, 5, 0], avail="bernoulli",
             runner_padding=True, per_weights=np.array([1.0, 0.5, 0.25, 0.8, 0.9]),
             argv=["--use_huber_loss", "--huber_delta", "1.0", "--use_per"])
    # 3. tiny dims, plain (non double-Q) targets
    run_case("qmix_tiny_nodouble", tiny, n_episodes=4, inds=[0, 1, 2, 3], avail="bernoulli", argv=["--use_double_q"])
    # 4. VDN (with the A-2 oracle patch)
    patch_vdn()
    run_case("vdn_tiny", tiny, n_episodes=4, inds=[2, 1, 0, 3], avail="bernoulli", vdn=True)
    # 5. KAT-A of SURVEY.md Appendix C: 3m dims, E=B=8, inds=arange(8); inputs regenerate from RandomState(0)
    k = run_case("qmix_3m_katA", DIMS["3m"], n_episodes=8, inds=np.arange(8), store_inputs=False)
    assert list(k["lengths"]) == [38, 30, 46, 38, 60, 35, 37, 57], k["lengths"]
    # 6. an odd-sized case (nothing a multiple of 4) to exercise the unaligned kernel paths
    odd = EnvDims("odd", 3, 7, 18, 54, 5)
    run_case("qmix_odd", odd, n_episodes=6, inds=[5, 1, 1, 2, 0, 4, 3], avail="bernoulli")
    # 7. previous action as a network input (config.py:81, QMixPolicy.py:29-33, qmix.py:123-124)
    run_case("qmix_tiny_prevact", tiny, n_episodes=5, inds=[4, 2, 0, 1], avail="bernoulli", argv=["--prev_act_inp"])
    # use_same_share_obs = False: per-agent centralized observations in the buffer, agent 0's feeds the mixer
    run_case("qmix_tiny_pershare", tiny, n_episodes=5, inds=[4, 1, 1, 0, 2], cap=6, pre_insert=3, avail="bernoulli", per_agent_share=True)


if __name__ == "__main__":
    main()
