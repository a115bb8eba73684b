"""-m gpu: every launch script the reference ships (offpolicy/scripts/train_*.sh) names a configuration -- algorithm, environment,
policy sharing, flags -- that must CONSTRUCT and TRAIN on the accelerated path, with no NotImplementedError on the way.

This is an acceptance matrix, not a parity test (parity of the same shapes is pinned by the step fixtures and the runner traces): for
each script the policies, the trainer and the replay buffer are built the way the reference's base runners build them
(runner/rnn/base_runner.py:60-150, runner/mlp/base_runner.py:60-150) from the script's command line and the observation / action spaces the
named environment reports, synthetic experience is inserted, and every policy is updated twice the way `batch_train` / `batch_train_q`
do (sample -> train_policy_on_batch -> target updates). Checked: finite losses and gradient norms, parameters that moved, targets that
followed. The command lines are quoted from the scripts (the GPU box has no /root/reference)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _spaces():
    from offpolicy_amd.utils.spaces import Discrete, MultiDiscrete
    # what the environments report: MPE scenarios at their default agent / landmark counts (envs/mpe/environment.py:60-110; the centralized
    # observation is all observations concatenated, mpe_runner.py:163-164); SMAC maps with the widths of offpolicy_amd.utils.synth.DIMS
    # (3m) and, for 3s5z_vs_3s6z, 8 agents / 15 actions / 136 observations / 230 state features
    return {
        "simple_spread": dict(obs=[18, 18, 18], act=[Discrete(5)] * 3),
        "simple_reference": dict(obs=[21, 21], act=[MultiDiscrete([[0, 4], [0, 9]])] * 2),
        "simple_speaker_listener": dict(obs=[3, 11], act=[Discrete(3), Discrete(5)]),
        "3m": dict(obs=[64] * 3, act=[Discrete(9)] * 3, state=48, episode_length=60),
        "3s5z_vs_3s6z": dict(obs=[136] * 8, act=[Discrete(15)] * 8, state=230, episode_length=170),
    }


# (script, algorithm_name, environment, share_policy, the flags of its command line that reach the update path)
SCRIPTS = [
    ("train_mpe_maddpg.sh", "maddpg", "simple_spread", True,
     dict(episode_length=25, actor_train_interval_step=1, tau=0.005, lr=7e-4, batch_size=1000, buffer_size=500000, use_reward_normalization=True)),
    ("train_mpe_matd3.sh", "matd3", "simple_reference", True, dict(episode_length=25, tau=0.005, lr=7e-4, batch_size=1000, buffer_size=500000)),
    # `--share_policy` is a store_false flag (config.py:61): the scripts that pass it run one policy per agent
    ("train_mpe_mqmix.sh", "mvdn", "simple_speaker_listener", False,
     dict(episode_length=25, use_soft_update=True, lr=7e-4, hard_update_interval_episode=500)),
    ("train_mpe_qmix.sh", "qmix", "simple_spread", True,
     dict(episode_length=25, batch_size=32, tau=0.005, lr=7e-4, hard_update_interval_episode=100, use_reward_normalization=True)),
    ("train_mpe_rmaddpg.sh", "rmaddpg", "simple_speaker_listener", False,
     dict(episode_length=25, actor_train_interval_step=1, tau=0.005, lr=7e-4, use_reward_normalization=True)),
    ("train_mpe_rmatd3.sh", "rmatd3", "simple_spread", True, dict(episode_length=25, tau=0.005, lr=7e-4, use_reward_normalization=True)),
    ("train_mpe_vdn.sh", "vdn", "simple_spread", True, dict(episode_length=25, use_soft_update=True, lr=7e-4, hard_update_interval_episode=200)),
    ("train_smac_qmix.sh", "qmix", "3s5z_vs_3s6z", True,
     dict(buffer_size=5000, lr=5e-4, batch_size=32, use_soft_update=True, hard_update_interval_episode=200, gain=1.0), "global_all_local"),
    ("train_smac_rmaddpg.sh", "rmaddpg", "3m", True, dict(lr=5e-4, buffer_size=5000, batch_size=32, actor_train_interval_step=1, tau=0.005)),
    ("train_smac_rmatd3.sh", "rmatd3", "3m", True, dict(lr=5e-4, buffer_size=5000, batch_size=32, tau=0.005, actor_train_interval_step=1)),
    ("train_smac_vdn.sh", "vdn", "3m", True, dict(buffer_size=5000, use_soft_update=True, hard_update_interval_episode=200)),
]


def _classes(algo):
    """The import table of the reference's base runners (runner/rnn/base_runner.py:62-85, runner/mlp/base_runner.py:62-85)."""
    if algo == "qmix":
        from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy as P
        from offpolicy_amd.algorithms.qmix.qmix import QMix as T
    elif algo == "vdn":
        from offpolicy_amd.algorithms.vdn.algorithm.VDNPolicy import VDNPolicy as P
        from offpolicy_amd.algorithms.vdn.vdn import VDN as T
    elif algo == "mqmix":
        from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy as P
        from offpolicy_amd.algorithms.mqmix.mqmix import M_QMix as T
    elif algo == "mvdn":
        from offpolicy_amd.algorithms.mvdn.algorithm.mVDNPolicy import M_VDNPolicy as P
        from offpolicy_amd.algorithms.mvdn.mvdn import M_VDN as T
    elif algo == "maddpg":
        from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy as P
        from offpolicy_amd.algorithms.maddpg.maddpg import MADDPG as T
    elif algo == "matd3":
        from offpolicy_amd.algorithms.matd3.algorithm.MATD3Policy import MATD3Policy as P
        from offpolicy_amd.algorithms.matd3.matd3 import MATD3 as T
    elif algo == "rmaddpg":
        from offpolicy_amd.algorithms.r_maddpg.algorithm.rMADDPGPolicy import R_MADDPGPolicy as P
        from offpolicy_amd.algorithms.r_maddpg.r_maddpg import R_MADDPG as T
    elif algo == "rmatd3":
        from offpolicy_amd.algorithms.r_matd3.algorithm.rMATD3Policy import R_MATD3Policy as P
        from offpolicy_amd.algorithms.r_matd3.r_matd3 import R_MATD3 as T
    else:
        raise AssertionError(algo)
    return P, T


def _onehot_actions(rng, space, shape):
    """Actions as the runners store them: one-hot (Discrete) or one one-hot block per sub-action (MultiDiscrete)."""
    from offpolicy_amd.utils.spaces import get_dim_from_space
    dim = get_dim_from_space(space)
    if np.ndim(dim) == 0:
        return np.eye(int(dim), dtype=np.float32)[rng.randint(0, int(dim), size=shape)]
    return np.concatenate([np.eye(int(k), dtype=np.float32)[rng.randint(0, int(k), size=shape)] for k in dim], axis=-1)


def _width(space):
    from offpolicy_amd.utils.spaces import get_dim_from_space
    return int(np.sum(get_dim_from_space(space)))


@pytest.mark.parametrize("entry", SCRIPTS, ids=[e[0] for e in SCRIPTS])
def test_launch_script_configuration_trains_on_the_accelerated_path(entry):
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    script, algo, env_name, share, flags = entry[:5]
    env = _spaces()[env_name]
    N = len(env["obs"])
    smac = "state" in env
    args = default_args(algorithm_name=algo, share_policy=share, **flags)
    if smac:
        args.episode_length = env["episode_length"]
    B = min(int(args.batch_size), 64)              # the script's batch size where it is small; capped so the matrix stays a few seconds
    cap = 96                                       # buffer capacity: a capacity, not a shape
    T = int(args.episode_length)
    state = env["state"] + (sum(env["obs"]) if len(entry) > 5 else 0) if smac else sum(env["obs"])     # MPE: all observations concatenated
    cent_act = sum(_width(a) for a in env["act"])
    recurrent = algo in ("qmix", "vdn", "rmaddpg", "rmatd3")
    dev = torch.device("cuda:0")
    if share:
        pinfo = {"policy_0": {"cent_obs_dim": state, "cent_act_dim": cent_act, "obs_space": [env["obs"][0]], "share_obs_space": [state],
                              "act_space": env["act"][0]}}
        mapping = lambda a: "policy_0"
    else:
        pinfo = {"policy_%d" % i: {"cent_obs_dim": state, "cent_act_dim": cent_act, "obs_space": [env["obs"][i]], "share_obs_space": [state],
                                   "act_space": env["act"][i]} for i in range(N)}
        mapping = lambda a: "policy_%d" % a
    pids = sorted(pinfo)
    pagents = {p: [a for a in range(N) if mapping(a) == p] for p in pids}
    Policy, Trainer = _classes(algo)
    torch.manual_seed(0)
    np.random.seed(0)
    policies = {p: Policy({"args": args, "device": dev}, pinfo[p]) for p in pids}
    kw = dict(episode_length=T) if recurrent else {}
    trainer = Trainer(args, N, policies, mapping, device=dev, **kw)
    rng = np.random.RandomState(1)
    f32 = np.float32
    E = 48
    if recurrent:
        buf = RecReplayBuffer(pinfo, pagents, cap, T, True, smac, bool(args.use_reward_normalization), device=dev)
        share_obs = rng.standard_normal((T + 1, E, 1, state)).astype(f32)
        lengths = rng.randint(max(2, T // 2), T + 1, size=E)
        dones_env = (np.arange(T)[:, None] >= lengths[None] - 1).astype(f32)[..., None]
        rew = rng.standard_normal((T, E, 1, 1)).astype(f32)
        d = {k: {} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")}
        for p in pids:
            n = len(pagents[p])
            d["obs"][p] = rng.standard_normal((T + 1, E, n, pinfo[p]["obs_space"][0])).astype(f32)
            d["share_obs"][p] = np.repeat(share_obs, n, axis=2)
            d["acts"][p] = _onehot_actions(rng, pinfo[p]["act_space"], (T, E, n))
            d["rewards"][p] = np.repeat(rew, n, axis=2)
            d["dones"][p] = np.repeat(dones_env[:, :, None], n, axis=2)
            d["dones_env"][p] = dones_env
            d["avail_acts"][p] = np.ones((T + 1, E, n, _width(pinfo[p]["act_space"])), f32) if smac else None
        buf.insert(E, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    else:
        buf = MlpReplayBuffer(pinfo, pagents, cap, True, False, bool(args.use_reward_normalization), device=dev)
        share_obs, nshare = rng.standard_normal((E, 1, state)).astype(f32), rng.standard_normal((E, 1, state)).astype(f32)
        dones_env = (rng.random_sample((E, 1)) < 0.1).astype(f32)
        rew = rng.standard_normal((E, 1, 1)).astype(f32)
        keys = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition", "avail_acts",
                "next_avail_acts")
        d = {k: {} for k in keys}
        for p in pids:
            n = len(pagents[p])
            D = pinfo[p]["obs_space"][0]
            d["obs"][p], d["next_obs"][p] = rng.standard_normal((E, n, D)).astype(f32), rng.standard_normal((E, n, D)).astype(f32)
            d["share_obs"][p], d["next_share_obs"][p] = np.repeat(share_obs, n, axis=1), np.repeat(nshare, n, axis=1)
            d["acts"][p] = _onehot_actions(rng, pinfo[p]["act_space"], (E, n))
            d["rewards"][p] = np.repeat(rew, n, axis=1)
            d["dones_env"][p] = dones_env
            d["dones"][p] = np.repeat(dones_env[:, None], n, axis=1)
            d["valid_transition"][p] = np.ones((E, n, 1), f32)
            d["avail_acts"][p], d["next_avail_acts"][p] = None, None
        buf.insert(E, *[d[k] for k in keys])

    def flat_params(pol):
        mods = [m for m in (getattr(pol, "q_network", None), getattr(pol, "actor", None), getattr(pol, "critic", None)) if m is not None]
        return torch.cat([v.detach().flatten() for m in mods for v in m.parameters()]).clone()

    before = {p: flat_params(policies[p]) for p in pids}
    q_learning = algo in ("qmix", "vdn", "mqmix", "mvdn")
    for _ in range(2):
        trainer.prep_training()
        if q_learning:          # batch_train_q: one sample, one update of all policies and the mixer
            info, _, _ = trainer.train_policy_on_batch(buf.sample(B))
            infos = [info]
        else:                   # batch_train: every policy in turn on its own sample
            infos = [trainer.train_policy_on_batch(p, buf.sample(B))[0] for p in pids]
        for info in infos:
            for k, v in info.items():
                if k.endswith("loss") or k.endswith("grad_norm"):
                    assert np.isfinite(float(v)), (script, k, v)
        if q_learning:
            trainer.soft_target_updates() if args.use_soft_update else trainer.hard_target_updates()
        else:
            for p in pids:
                policies[p].soft_target_updates()
    torch.cuda.synchronize()
    for p in pids:
        after = flat_params(policies[p])
        assert torch.isfinite(after).all() and (after != before[p]).float().mean() > 0.3, (script, p)
