"""Record what the REFERENCE's MLP MPE runner does for MVDN with ONE POLICY PER AGENT on a speaker / listener pair
(tests/golden/runner_trace_mvdn_multi.npz) -- the configuration of scripts/train_mpe_mqmix.sh (algo mvdn, simple_speaker_listener,
share_policy off; upstream the script stops in argparse on a misspelt flag and M_VDNMixer.forward lacks the state argument, SURVEY A-1:
run here with the flag dropped and the documented one-line oracle patch).

TEST INFRASTRUCTURE ONLY. Run in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_runner_trace_mvdn.py

The reference's own `MPERunner` (offpolicy/runner/mlp/mpe_runner.py `separated_collect_rollout` + base_runner.py `batch_train_q`) on a
deterministic stub environment whose agents differ in observation width and action count: two M_QMixPolicy constructions, the M_QMix(vdn)
trainer, warm-up with random actions, two run() cycles -- per environment step every agent queried through its own policy
(`get_actions(explore)`), one 12-argument two-policy `buffer.insert`, every second step `batch_train_q`: per policy `buffer.sample` ->
`train_policy_on_batch(sample, use_same_share_obs)` (ALL policies' networks under the VDN sum), then `soft_target_updates`. Recording
subclasses log every call with arguments, RNG states and return values; tests/test_gpu_runner_trace.py replays them on the engine."""
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
import offpolicy.algorithms.mqmix.algorithm.mQMixPolicy as ref_policy_mod  # noqa: E402
import offpolicy.algorithms.mqmix.mqmix as ref_trainer_mod  # noqa: E402
from oracle.make_golden_mlp import patch_mvdn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "runner_trace_mvdn_multi.npz")
SHAPES = [(3, 3), (11, 5)]          # (observation width, actions): speaker, listener
N, T = len(SHAPES), 5
S = sum(d for d, _ in SHAPES)
PIDS = ["policy_%d" % i for i in range(N)]
INS = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition")
LOG = []
STORE = {}


def _np(x):
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
    """One environment (num_envs = 1) with ragged per-agent observations [[obs_0 (3,), obs_1 (11,)]], shared reward, all agents done
    every T-th step. Its own RandomState: independent of the global RNGs."""
    num_envs = 1

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.t = 0

    def _obs(self):
        return [[self.rng.standard_normal(d).astype(np.float32) for d, _ in SHAPES]]

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, env_acts):
        self.t += 1
        r = float(sum(np.asarray(a).argmax() for a in env_acts[0])) * 0.1 + float(self.rng.standard_normal()) * 0.05
        return self._obs(), np.full((1, N, 1), r, np.float32), np.full((1, N), self.t % T == 0, dtype=bool), [[{} for _ in range(N)]]


class RecBuffer(ref_buffer_mod.MlpReplayBuffer):
    def insert(self, num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env, valid_transition,
               avail_acts, next_avail_acts):
        out = super().insert(num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env, valid_transition,
                             avail_acts, next_avail_acts)
        vals = dict(obs=obs, share_obs=share_obs, acts=acts, rewards=rewards, next_obs=next_obs, next_share_obs=next_share_obs, dones=dones,
                    dones_env=dones_env, valid_transition=valid_transition)
        log_call("buffer.insert", dict(n=np.array(num_insert_steps), **{"%s/%s" % (p, k): np.asarray(vals[k][p]) for p in PIDS for k in INS}),
                 dict(idx_range=out))
        return out

    def sample(self, batch_size):
        st = rng_state()
        out = super().sample(batch_size)
        log_call("buffer.sample", dict(batch_size=np.array(batch_size), **st), {"%s/%s" % (p, k): out[i][p] for p in PIDS for i, k in enumerate(INS)})
        return out


class RecPolicy(ref_policy_mod.M_QMixPolicy):
    _count = [0]

    def __init__(self, config, policy_config, *a, **k):
        st = rng_state()
        super().__init__(config, policy_config, *a, **k)
        self._pid = RecPolicy._count[0]
        RecPolicy._count[0] += 1
        log_call("policy.__init__", dict(pid=np.array(self._pid), **st), {"sd/" + kk: v for kk, v in self.q_network.state_dict().items()})

    def get_actions(self, obs, available_actions=None, t_env=None, explore=False):
        st = rng_state()
        out = super().get_actions(obs, available_actions, t_env, explore)
        if hasattr(self, "_pid") and not getattr(RecPolicy, "_in_train", False):
            log_call("policy.get_actions", dict(pid=np.array(self._pid), obs=obs, t_env=None if t_env is None else np.array(t_env),
                                                explore=np.array(bool(explore)), **st), dict(actions=out[0]))
        return out

    def get_random_actions(self, obs, available_actions=None):
        st = rng_state()
        out = super().get_random_actions(obs, available_actions)
        log_call("policy.get_random_actions", dict(pid=np.array(self._pid), obs=obs, **st), dict(actions=out))
        return out


class RecTrainer(ref_trainer_mod.M_QMix):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        log_call("trainer.__init__", {}, {})

    def train_policy_on_batch(self, batch, use_same_share_obs):
        st = rng_state()
        RecPolicy._in_train = True           # (target policies are deep copies of the recording policies)
        try:
            out = super().train_policy_on_batch(batch, use_same_share_obs)
        finally:
            RecPolicy._in_train = False
        info = out[0]
        log_call("trainer.train_policy_on_batch", dict(use_same_share_obs=np.array(bool(use_same_share_obs)), **st),
                 {k: info[k] for k in ("loss", "grad_norm", "Q_tot")})
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
    patch_mvdn()
    ref_buffer_mod.MlpReplayBuffer = RecBuffer
    ref_policy_mod.M_QMixPolicy = RecPolicy
    ref_trainer_mod.M_QMix = RecTrainer
    from offpolicy.runner.mlp.mpe_runner import MPERunner       # binds the recording classes
    args = reference_args(["--algorithm_name", "mvdn", "--env_name", "MPE", "--batch_size", "8", "--buffer_size", "64",
                           "--num_random_episodes", "2", "--episode_length", str(T), "--epsilon_anneal_time", "40", "--lr", "1e-3",
                           "--train_interval", "2"],
                          scenario_name="stub", experiment_name="trace", use_wandb=False, use_eval=False, save_interval=10 ** 9,
                          log_interval=10 ** 9, share_policy=False)
    pinfo = {p: {"cent_obs_dim": S, "cent_act_dim": sum(a for _, a in SHAPES), "obs_space": [d], "share_obs_space": [S], "act_space": Discrete(a)}
             for p, (d, a) in zip(PIDS, SHAPES)}
    torch.manual_seed(3)
    np.random.seed(3)
    with tempfile.TemporaryDirectory() as tmp:
        config = {"args": args, "policy_info": pinfo, "policy_mapping_fn": lambda a: "policy_%d" % a, "env": StubEnv(1), "eval_env": StubEnv(2),
                  "num_agents": N, "device": torch.device("cpu"), "use_same_share_obs": True, "run_dir": Path(tmp)}
        runner = MPERunner(config)          # constructor + warm-up
        for _ in range(2):
            runner.run()
        final = {}
        for p in PIDS:
            final.update({"%s/live/%s" % (p, k): v for k, v in runner.policies[p].q_network.state_dict().items()})
            final.update({"%s/target/%s" % (p, k): v for k, v in runner.trainer.target_policies[p].q_network.state_dict().items()})
        log_call("runner.final_state", {}, final)
    STORE["calls"] = np.array(LOG)
    STORE["shapes"] = np.array(SHAPES)
    STORE["dims"] = np.array([N, S, T])
    STORE["hp"] = np.array([args.batch_size, args.buffer_size, args.lr, args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time],
                           dtype=np.float64)
    np.savez_compressed(OUT, **STORE)
    from collections import Counter
    print(len(LOG), "calls:", dict(Counter(LOG)))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
