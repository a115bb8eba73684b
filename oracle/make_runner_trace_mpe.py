"""Record what the REFERENCE's MPE runner does to its buffer / policy / trainer objects with ONE POLICY PER AGENT
(tests/golden/runner_trace_rmaddpg_multi.npz) -- scripts/train_mpe_rmaddpg.sh's configuration (share_policy = False).

TEST INFRASTRUCTURE ONLY. Run in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_runner_trace_mpe.py

Same method as oracle/make_runner_trace.py: the reference's own `MPERunner` (offpolicy/runner/rnn/mpe_runner.py +
base_runner.py) is constructed on a small deterministic stub environment and driven through its normal life cycle --
constructor (three R_MADDPGPolicy objects, the R_MADDPG trainer, the three-policy buffer), warm-up episodes with random
actions (`separated_collect_rollout`: every agent queried through its own policy), then `run()` a few times (exploring
rollout -> buffer.insert -> batch_train: per policy buffer.sample -> shared_train_policy_on_batch -> soft updates of every
policy). The runner runs on recording subclasses of the reference's classes, which log every call with its arguments, the
numpy / torch RNG state before it, and what the reference returned; tests/test_gpu_runner_trace.py replays the sequence
against the engine."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference, reference_args  # noqa: E402

load_reference()
from gym.spaces import Discrete  # noqa: E402
import offpolicy.utils.rec_buffer as ref_rec_buffer  # noqa: E402
import offpolicy.algorithms.r_maddpg.algorithm.rMADDPGPolicy as ref_policy_mod  # noqa: E402
import offpolicy.algorithms.r_maddpg.r_maddpg as ref_trainer_mod  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "runner_trace_rmaddpg_multi.npz")
N, A, D, T = 3, 5, 6, 5
AS, DS = [A] * N, [D] * N
if os.environ.get("OPE_TRACE") == "sl":
    # the shapes of MPE simple_speaker_listener, the scenario scripts/train_mpe_rmaddpg.sh names: agent 0 (speaker) sees 3 numbers and has 3
    # actions, agent 1 (listener) sees 11 and has 5; PYTHONDONTWRITEBYTECODE=1 OPE_TRACE=sl python oracle/make_runner_trace_mpe.py
    OUT = os.path.join(ROOT, "tests", "golden", "runner_trace_rmaddpg_sl.npz")
    N, T = 2, 5
    AS, DS = [3, 5], [3, 11]
    A, D = max(AS), max(DS)
S = sum(DS)        # the MPE runner's shared observation = all agents' observations concatenated (mpe_runner.py:163-164)
PIDS = ["policy_%d" % i for i in range(N)]
KEYS = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env")
LOG = []
STORE = {}


def _np(x):
    if x is None:
        return None
    if torch.is_tensor(x):
        return x.detach().cpu().numpy().copy()
    return np.array(x, copy=True)


def log_call(name, inputs, outputs):
    pre = "c%03d/" % len(LOG)
    LOG.append(name)
    for k, v in inputs.items():
        if v is not None:
            STORE[pre + "in/" + k] = _np(v)
    for k, v in outputs.items():
        if v is not None:
            STORE[pre + "out/" + k] = _np(v)


def rng_state():
    st = np.random.get_state()
    return {"np_keys": st[1].copy(), "np_pos": np.array([st[2], st[3]], dtype=np.int64), "np_gauss": np.array([st[4]]),
            "torch": torch.get_rng_state().numpy().copy()}


class StubEnv(object):
    """One MPE-like environment (num_envs = 1): obs [1, N, D]; rewards [1, N, 1] (shared); dones [1, N], all true at a step drawn
    per episode in [3, T]. Its own RandomState: independent of the global RNGs."""

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.t = 0

    def _obs(self):
        if len(set(DS)) == 1:
            return self.rng.standard_normal((1, N, D)).astype(np.float32)
        obs = np.empty((1, N), dtype=object)      # agents of different observation widths: an object array, as the MPE wrappers return
        for i in range(N):
            obs[0, i] = self.rng.standard_normal(DS[i]).astype(np.float32)
        return obs

    def reset(self):
        self.t = 0
        self.end = self.rng.randint(3, T + 1)
        return self._obs()

    def step(self, env_acts):
        self.t += 1
        r = float(sum(int(np.argmax(a)) for a in env_acts[0])) * 0.1 + float(self.rng.standard_normal()) * 0.05
        rewards = np.full((1, N, 1), r, np.float32)
        dones = np.full((1, N), self.t >= self.end, dtype=bool)
        return self._obs(), rewards, dones, [[{} for _ in range(N)]]


class RecBuffer(ref_rec_buffer.RecReplayBuffer):
    def insert(self, num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts):
        out = super().insert(num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts)
        vals = dict(obs=obs, share_obs=share_obs, acts=acts, rewards=rewards, dones=dones, dones_env=dones_env)
        log_call("buffer.insert", dict(n=np.array(num_insert_episodes), **{p + "/" + k: vals[k][p] for p in PIDS for k in KEYS}),
                 dict(idx_range=out))
        return out

    def sample(self, batch_size):
        st = rng_state()
        out = super().sample(batch_size)
        log_call("buffer.sample", dict(batch_size=np.array(batch_size), **st),
                 {p + "/" + k: out[i][p] for p in PIDS for i, k in enumerate(KEYS)})
        return out


class RecPolicy(ref_policy_mod.R_MADDPGPolicy):
    _count = [0]

    def __init__(self, config, policy_config, *a, **k):
        st = rng_state()
        super().__init__(config, policy_config, *a, **k)
        self._pid = PIDS[RecPolicy._count[0]]
        RecPolicy._count[0] += 1
        out = {}
        for grp, mod in (("actor", self.actor), ("critic", self.critic)):
            out.update({"sd/%s/%s" % (grp, kk): v for kk, v in mod.state_dict().items()})
        log_call("policy.__init__", dict(pid=np.array(PIDS.index(self._pid)), **st), out)

    def get_actions(self, obs, prev_actions, rnn_states, available_actions=None, t_env=None, explore=False, use_target=False, use_gumbel=False):
        st = rng_state()
        out = super().get_actions(obs, prev_actions, rnn_states, available_actions, t_env, explore, use_target, use_gumbel)
        if not use_target and not use_gumbel and not torch.is_grad_enabled():      # the runner's rollout calls (trainer calls are internal)
            log_call("policy.get_actions", dict(pid=np.array(PIDS.index(self._pid)), obs=obs, prev_actions=prev_actions, rnn_states=rnn_states,
                                                t_env=None if t_env is None else np.array(t_env), explore=np.array(bool(explore)), **st),
                     dict(actions=out[0], rnn_states=out[1]))
        return out

    def get_random_actions(self, obs, available_actions=None):
        st = rng_state()
        out = super().get_random_actions(obs, available_actions)
        log_call("policy.get_random_actions", dict(pid=np.array(PIDS.index(self._pid)), obs=obs, **st), dict(actions=out))
        return out

    def soft_target_updates(self):
        super().soft_target_updates()
        log_call("policy.soft_target_updates", dict(pid=np.array(PIDS.index(self._pid))), {})


class RecTrainer(ref_trainer_mod.R_MADDPG):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        log_call("trainer.__init__", {}, {})

    def shared_train_policy_on_batch(self, update_policy_id, batch):
        st = rng_state()
        out = super().shared_train_policy_on_batch(update_policy_id, batch)
        info = out[0]
        log_call("trainer.shared_train_policy_on_batch", dict(pid=np.array(PIDS.index(update_policy_id)), **st),
                 {k: info[k] for k in ("critic_loss", "critic_grad_norm", "actor_loss", "actor_grad_norm") if k in info})
        return out

    def prep_rollout(self):
        super().prep_rollout()
        log_call("trainer.prep_rollout", {}, {})

    def prep_training(self):
        super().prep_training()
        log_call("trainer.prep_training", {}, {})


def main():
    ref_rec_buffer.RecReplayBuffer = RecBuffer
    ref_policy_mod.R_MADDPGPolicy = RecPolicy
    ref_trainer_mod.R_MADDPG = RecTrainer
    from offpolicy.runner.rnn.mpe_runner import MPERunner       # binds the recording classes
    args = reference_args(["--algorithm_name", "rmaddpg", "--env_name", "MPE", "--batch_size", "4", "--buffer_size", "10",
                           "--num_random_episodes", "4", "--episode_length", str(T), "--epsilon_anneal_time", "40", "--lr", "1e-3",
                           "--share_policy", "--actor_train_interval_step", "1"],
                          scenario_name="stub", experiment_name="trace", use_wandb=False, use_eval=False, save_interval=10 ** 9, log_interval=10 ** 9)
    assert args.share_policy is False
    pinfo = {p: {"cent_obs_dim": S, "cent_act_dim": sum(AS), "obs_space": [d], "share_obs_space": [S], "act_space": Discrete(a)}
             for p, d, a in zip(PIDS, DS, AS)}
    torch.manual_seed(3)
    np.random.seed(3)
    with tempfile.TemporaryDirectory() as tmp:
        config = {"args": args, "policy_info": pinfo, "policy_mapping_fn": lambda a: "policy_%d" % a, "env": StubEnv(1), "eval_env": StubEnv(2),
                  "num_agents": N, "device": torch.device("cpu"), "use_same_share_obs": True, "run_dir": Path(tmp)}
        runner = MPERunner(config)          # constructor + warm-up
        for _ in range(3):
            runner.run()
        final = {}
        for p in PIDS:
            pol = runner.policies[p]
            for grp, mod in (("actor", pol.actor), ("critic", pol.critic), ("target_actor", pol.target_actor), ("target_critic", pol.target_critic)):
                final.update({"%s/%s/%s" % (p, grp, k): v for k, v in mod.state_dict().items()})
        log_call("runner.final_state", {}, final)
    STORE["calls"] = np.array(LOG)
    STORE["dims"] = np.array([N, A, D, S, T])
    if len(set(DS)) > 1 or len(set(AS)) > 1:
        STORE["obs_dims"], STORE["act_dims"] = np.array(DS), np.array(AS)
    STORE["hp"] = np.array([args.batch_size, args.buffer_size, args.lr, args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time],
                           dtype=np.float64)
    np.savez_compressed(OUT, **STORE)
    from collections import Counter
    print(len(LOG), "calls:", dict(Counter(LOG)))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
