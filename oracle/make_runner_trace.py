"""Record what the REFERENCE's runner does to its buffer / policy / trainer objects (tests/golden/runner_trace_qmix.npz).

TEST INFRASTRUCTURE ONLY. Run in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_runner_trace.py

The reference's own `SMACRunner` (offpolicy/runner/rnn/smac_runner.py + base_runner.py) is constructed on a small
deterministic stub environment and driven through its normal life cycle -- constructor (policies, trainer, buffer), warm-up
episodes with random actions, then `run()` (collect_rollout with epsilon-greedy exploration -> buffer.insert -> batch_train_q:
buffer.sample -> trainer.train_policy_on_batch -> soft_target_updates) a few times, and `save_q()`. The runner runs on
recording subclasses of the reference's RecReplayBuffer / QMixPolicy / QMix, which log every call the runner makes: method,
arguments, the numpy / torch RNG state before the call, and what the reference returned. tests/test_gpu_runner_trace.py
replays that exact call sequence, with those arguments and RNG states, against the engine's classes on the GPU and compares
every return value: the drop-in surface exercised by the real runner rather than by hand-written calls. The reference runner
cannot itself be run against the engine: it only exists in the build container (no GPU), the engine only runs on the GPU box
(no reference tree)."""
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
import offpolicy.algorithms.qmix.algorithm.QMixPolicy as ref_policy_mod  # noqa: E402
import offpolicy.algorithms.qmix.qmix as ref_qmix_mod  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "runner_trace_qmix.npz")
N, A, D, S, T = 3, 5, 10, 12, 6
LOG = []          # (name, {key: array})
STORE = {}


def _np(x):
    if x is None:
        return None
    if torch.is_tensor(x):
        return x.detach().cpu().numpy().copy()
    return np.array(x, copy=True)


def log_call(name, inputs, outputs, rng=True):
    i = len(LOG)
    pre = "c%03d/" % i
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
    """One SMAC-like environment (num_envs = 1): obs [1, N, D], share_obs [1, N, S], avail [1, N, A]; the episode ends at a
    step drawn per episode in [3, T]; rewards depend on the joint action. Its own RandomState: independent of the global RNGs."""

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.t = 0

    def _obs(self):
        obs = self.rng.standard_normal((1, N, D)).astype(np.float32)
        st = self.rng.standard_normal((1, 1, S)).astype(np.float32).repeat(N, axis=1)
        av = (self.rng.random_sample((1, N, A)) < 0.7).astype(np.float32)
        av[..., 0] = 1.0
        return obs, st, av

    def reset(self):
        self.t = 0
        self.end = self.rng.randint(3, T + 1)
        return self._obs()

    def step(self, env_acts):
        self.t += 1
        acts = np.asarray(env_acts[0])
        r = float(acts.argmax(-1).sum()) * 0.1 + float(self.rng.standard_normal()) * 0.05
        rewards = np.full((1, N, 1), r, np.float32)
        done = self.t >= self.end
        dones = np.full((1, N, 1), done, dtype=bool)
        infos = [[{"won": bool(done and r > 0.5)} for _ in range(N)]]
        obs, st, av = self._obs()
        return obs, st, rewards, dones, infos, av


# ---- recording subclasses ------------------------------------------------------------------------------------------------
class RecBuffer(ref_rec_buffer.RecReplayBuffer):
    def insert(self, num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts):
        out = super().insert(num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts)
        p = "policy_0"
        log_call("buffer.insert", dict(n=np.array(num_insert_episodes), obs=obs[p], share_obs=share_obs[p], acts=acts[p], rewards=rewards[p],
                                       dones=dones[p], dones_env=dones_env[p], avail_acts=avail_acts[p]), dict(idx_range=out))
        return out

    def sample(self, batch_size):
        st = rng_state()
        out = super().sample(batch_size)
        p = "policy_0"
        keys = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")
        log_call("buffer.sample", dict(batch_size=np.array(batch_size), **st), {k: out[i][p] for i, k in enumerate(keys)})
        return out


class RecPolicy(ref_policy_mod.QMixPolicy):
    def __init__(self, config, policy_config, train=True):
        st = rng_state()
        super().__init__(config, policy_config, train)
        log_call("policy.__init__", st, {"sd/" + k: v for k, v in self.q_network.state_dict().items()})

    def get_actions(self, obs, prev_actions, rnn_states, available_actions=None, t_env=None, explore=False):
        st = rng_state()
        out = super().get_actions(obs, prev_actions, rnn_states, available_actions, t_env, explore)
        log_call("policy.get_actions", dict(obs=obs, prev_actions=prev_actions, rnn_states=rnn_states, available_actions=available_actions,
                                            t_env=None if t_env is None else np.array(t_env), explore=np.array(bool(explore)), **st),
                 dict(actions=out[0], rnn_states=out[1], greedy_Qs=out[2]))
        return out

    def get_random_actions(self, obs, available_actions=None):
        st = rng_state()
        out = super().get_random_actions(obs, available_actions)
        log_call("policy.get_random_actions", dict(obs=obs, available_actions=available_actions, **st), dict(actions=out))
        return out


class RecTrainer(ref_qmix_mod.QMix):
    def __init__(self, *a, **k):
        st = rng_state()
        super().__init__(*a, **k)
        log_call("trainer.__init__", st, {"sd/" + kk: v for kk, v in self.mixer.state_dict().items()})

    def train_policy_on_batch(self, batch, update_policy_id=None):
        out = super().train_policy_on_batch(batch, update_policy_id)
        info = out[0]
        log_call("trainer.train_policy_on_batch", {}, dict(loss=info["loss"], grad_norm=info["grad_norm"], Q_tot=info["Q_tot"]))
        return out

    def soft_target_updates(self):
        super().soft_target_updates()
        log_call("trainer.soft_target_updates", {}, {})

    def prep_rollout(self):
        super().prep_rollout()
        log_call("trainer.prep_rollout", {}, {})

    def prep_training(self):
        super().prep_training()
        log_call("trainer.prep_training", {}, {})


def main():
    ref_rec_buffer.RecReplayBuffer = RecBuffer
    ref_policy_mod.QMixPolicy = RecPolicy
    ref_qmix_mod.QMix = RecTrainer
    from offpolicy.runner.rnn.smac_runner import SMACRunner       # binds the recording classes
    args = reference_args(["--algorithm_name", "qmix", "--env_name", "StarCraft2", "--batch_size", "4", "--buffer_size", "8",
                           "--num_random_episodes", "4", "--episode_length", str(T), "--epsilon_anneal_time", "40", "--lr", "1e-3"],
                          map_name="stub", experiment_name="trace", use_wandb=False, use_eval=False, save_interval=10 ** 9, log_interval=10 ** 9)
    pinfo = {"policy_0": {"cent_obs_dim": S, "cent_act_dim": A * N, "obs_space": [D], "share_obs_space": [S], "act_space": Discrete(A)}}
    torch.manual_seed(3)
    np.random.seed(3)
    with tempfile.TemporaryDirectory() as tmp:
        config = {"args": args, "policy_info": pinfo, "policy_mapping_fn": lambda a: "policy_0", "env": StubEnv(1), "eval_env": StubEnv(2),
                  "num_agents": N, "device": torch.device("cpu"), "use_same_share_obs": True, "use_available_actions": True,
                  "run_dir": Path(tmp), "buffer_length": T}
        runner = SMACRunner(config)          # constructor + warm-up
        for _ in range(3):
            runner.run()                     # collect (epsilon-greedy) -> insert -> sample -> train -> soft update
        runner.saver()                       # save_q: q_network.pt + mixer.pt
        q_sd = torch.load(os.path.join(runner.save_dir, "policy_0", "q_network.pt"))
        m_sd = torch.load(os.path.join(runner.save_dir, "mixer.pt"))
        log_call("runner.save_q", {}, dict(**{"q/" + k: v for k, v in q_sd.items()}, **{"m/" + k: v for k, v in m_sd.items()}))
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
