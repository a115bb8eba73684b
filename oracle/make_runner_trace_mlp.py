"""Record what the REFERENCE's MLP MPE runner does to its buffer / policy / trainer objects for MADDPG-MLP with one shared
policy (tests/golden/runner_trace_maddpg.npz) -- the configuration of scripts/train_mpe_maddpg.sh (BASELINE config 2's family).

TEST INFRASTRUCTURE ONLY. Run in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_runner_trace_mlp.py

Same method as oracle/make_runner_trace.py: the reference's own `MPERunner` (offpolicy/runner/mlp/mpe_runner.py +
base_runner.py) on a small deterministic stub environment: constructor (MADDPGPolicy, MADDPG, MlpReplayBuffer), warm-up with
random actions, then `run()` a few times -- per environment step `get_actions(explore=True)` -> 12-argument `buffer.insert` of
one transition -> every `train_interval` steps `batch_train` (buffer.sample -> shared_train_policy_on_batch ->
policy.soft_target_updates). Recording subclasses log every call with arguments, RNG states and return values;
tests/test_gpu_runner_trace.py replays the sequence against the engine."""
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
import offpolicy.utils.mlp_buffer as ref_buffer_mod  # noqa: E402
import offpolicy.algorithms.maddpg.algorithm.MADDPGPolicy as ref_policy_mod  # noqa: E402
import offpolicy.algorithms.maddpg.maddpg as ref_trainer_mod  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "runner_trace_maddpg.npz")
N, A, D, T = 3, 5, 6, 5
S = N * D
P = "policy_0"
INS = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition")
LOG = []
STORE = {}
IN_TRAIN = [False]         # the trainer's own get_actions calls are internal to the update: not logged


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
    """One MPE-like vectorised environment (num_envs = 1): obs [1, N, D], rewards [1, N, 1] (shared), dones [1, N, 1] -- all true
    every T-th step. Its own RandomState: independent of the global RNGs."""
    num_envs = 1

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.t = 0

    def _obs(self):
        return self.rng.standard_normal((1, N, D)).astype(np.float32)

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, env_acts):
        self.t += 1
        acts = np.asarray(env_acts[0])
        r = float(acts.argmax(-1).sum()) * 0.1 + float(self.rng.standard_normal()) * 0.05
        return self._obs(), np.full((1, N, 1), r, np.float32), np.full((1, N, 1), self.t % T == 0, dtype=bool), [[{} for _ in range(N)]]


class RecBuffer(ref_buffer_mod.MlpReplayBuffer):
    def insert(self, num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env, valid_transition,
               avail_acts, next_avail_acts):
        out = super().insert(num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env, valid_transition,
                             avail_acts, next_avail_acts)
        vals = dict(obs=obs, share_obs=share_obs, acts=acts, rewards=rewards, next_obs=next_obs, next_share_obs=next_share_obs, dones=dones,
                    dones_env=dones_env, valid_transition=valid_transition)
        log_call("buffer.insert", dict(n=np.array(num_insert_steps), **{k: np.asarray(vals[k][P]) for k in INS}), dict(idx_range=out))
        return out

    def sample(self, batch_size):
        st = rng_state()
        out = super().sample(batch_size)
        log_call("buffer.sample", dict(batch_size=np.array(batch_size), **st), {k: out[i][P] for i, k in enumerate(INS)})
        return out


class RecPolicy(ref_policy_mod.MADDPGPolicy):
    def __init__(self, config, policy_config, *a, **k):
        st = rng_state()
        super().__init__(config, policy_config, *a, **k)
        out = {}
        for grp, mod in (("actor", self.actor), ("critic", self.critic)):
            out.update({"sd/%s/%s" % (grp, kk): v for kk, v in mod.state_dict().items()})
        for kq, head in enumerate(self.critic.q_outs):        # plain Python list upstream (A-4): not in state_dict()
            out["head/%d/weight" % kq], out["head/%d/bias" % kq] = head.weight, head.bias
        for kq, head in enumerate(self.target_critic.q_outs):
            out["thead/%d/weight" % kq], out["thead/%d/bias" % kq] = head.weight, head.bias
        log_call("policy.__init__", st, out)

    def get_actions(self, obs, available_actions=None, t_env=None, explore=False, use_target=False, use_gumbel=False):
        st = rng_state()
        out = super().get_actions(obs, available_actions, t_env, explore, use_target, use_gumbel)
        if not IN_TRAIN[0]:      # the runner's rollout calls
            log_call("policy.get_actions", dict(obs=obs, t_env=None if t_env is None else np.array(t_env), explore=np.array(bool(explore)), **st),
                     dict(actions=out[0]))
        return out

    def get_random_actions(self, obs, available_actions=None):
        st = rng_state()
        out = super().get_random_actions(obs, available_actions)
        log_call("policy.get_random_actions", dict(obs=obs, **st), dict(actions=out))
        return out

    def soft_target_updates(self):
        super().soft_target_updates()
        log_call("policy.soft_target_updates", {}, {})


class RecTrainer(ref_trainer_mod.MADDPG):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        log_call("trainer.__init__", {}, {})

    def shared_train_policy_on_batch(self, update_policy_id, batch):
        st = rng_state()
        IN_TRAIN[0] = True
        try:
            out = super().shared_train_policy_on_batch(update_policy_id, batch)
        finally:
            IN_TRAIN[0] = False
        info = out[0]
        log_call("trainer.shared_train_policy_on_batch", st,
                 {k: info[k] for k in ("critic_loss", "critic_grad_norm", "actor_loss", "actor_grad_norm") if k in info})
        return out

    def prep_rollout(self):
        super().prep_rollout()
        log_call("trainer.prep_rollout", {}, {})

    def prep_training(self):
        super().prep_training()
        log_call("trainer.prep_training", {}, {})


def main():
    ref_buffer_mod.MlpReplayBuffer = RecBuffer
    ref_policy_mod.MADDPGPolicy = RecPolicy
    ref_trainer_mod.MADDPG = RecTrainer
    from offpolicy.runner.mlp.mpe_runner import MPERunner       # binds the recording classes
    args = reference_args(["--algorithm_name", "maddpg", "--env_name", "MPE", "--batch_size", "8", "--buffer_size", "64",
                           "--num_random_episodes", "2", "--episode_length", str(T), "--epsilon_anneal_time", "40", "--lr", "1e-3",
                           "--train_interval", "2"],
                          scenario_name="stub", experiment_name="trace", use_wandb=False, use_eval=False, save_interval=10 ** 9, log_interval=10 ** 9)
    pinfo = {P: {"cent_obs_dim": S, "cent_act_dim": A * N, "obs_space": [D], "share_obs_space": [S], "act_space": Discrete(A)}}
    torch.manual_seed(3)
    np.random.seed(3)
    with tempfile.TemporaryDirectory() as tmp:
        config = {"args": args, "policy_info": pinfo, "policy_mapping_fn": lambda a: P, "env": StubEnv(1), "eval_env": StubEnv(2),
                  "num_agents": N, "device": torch.device("cpu"), "use_same_share_obs": True, "run_dir": Path(tmp)}
        runner = MPERunner(config)          # constructor + warm-up
        for _ in range(2):
            runner.run()
        pol = runner.policies[P]
        final = {}
        for grp, mod in (("actor", pol.actor), ("critic", pol.critic), ("target_actor", pol.target_actor), ("target_critic", pol.target_critic)):
            final.update({"%s/%s" % (grp, k): v for k, v in mod.state_dict().items()})
        log_call("runner.final_state", {}, final)
    STORE["calls"] = np.array(LOG)
    STORE["dims"] = np.array([N, A, D, S, T])
    STORE["hp"] = np.array([args.batch_size, args.buffer_size, args.lr, args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time],
                           dtype=np.float64)
    np.savez_compressed(OUT, **STORE)
    from collections import Counter
    print(len(LOG), "calls:", dict(Counter(LOG)))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
