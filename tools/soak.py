"""Soak run on the GPU: the reference's training loop shape (insert new episodes / transitions, prioritized sample, train, update priorities,
target update) for many steps, watching for a non-finite loss, drifting device memory and stale-state mistakes that short fixtures cannot show.
    python tools/soak.py [steps]          # QMIX-RNN 3s5z with PER (device trees), then MADDPG-MLP simple_spread with PER"""
import contextlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd.config import default_args  # noqa: E402
from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_episodes, as_policy_dicts  # noqa: E402


def qmix(steps, dev):
    from offpolicy_amd.utils.rec_buffer import PrioritizedRecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    dims = DIMS["3s5z"]
    args = default_args(use_per=True)
    pinfo = policy_info_for(dims)
    torch.manual_seed(1)
    np.random.seed(1)
    policy = QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
    with contextlib.redirect_stdout(sys.stderr):
        trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda x: "policy_0", device=dev, episode_length=dims.episode_length)
    cap = 600
    buf = PrioritizedRecReplayBuffer(args.per_alpha, pinfo, {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, True, True, device=dev,
                                     device_tree=True)
    rng = np.random.RandomState(0)
    pool = as_policy_dicts(synth_episodes(rng, 64, dims, avail="bernoulli"))
    keys = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")

    def insert(n):
        sel = rng.randint(0, 64, size=n)
        part = {k: {"policy_0": pool[k]["policy_0"][:, sel]} for k in keys}      # (insert layout is time-major: [T(+1), episodes, ...])
        buf.insert(n, *[part[k] for k in keys])
    for _ in range(8):
        insert(64)
    mem0, losses = None, []
    for st in range(steps):
        if st % 25 == 0:
            insert(8)                                          # the ring wraps around several times during the run
        beta = min(1.0, 0.4 + st / steps)
        sample = buf.sample(32, beta, "policy_0")
        info, prio, idxes = trainer.train_policy_on_batch(sample)
        buf.update_priorities(idxes, prio, "policy_0")
        trainer.soft_target_updates()
        if st % 500 == 499:
            loss = float(info["loss"])
            assert np.isfinite(loss) and np.isfinite(float(info["grad_norm"])), (st, loss)
            losses.append(loss)
            mem = torch.cuda.memory_allocated(dev)
            mem0 = mem if mem0 is None else mem0
            assert mem <= mem0 + 64 * 2 ** 20, "device memory grows: %d -> %d" % (mem0, mem)
    print("qmix 3s5z + PER: %d steps, loss every 500: %s, memory %.1f MB" % (steps, " ".join("%.4f" % x for x in losses), mem0 / 2 ** 20))


def maddpg(steps, dev):
    from offpolicy_amd.utils.mlp_buffer import PrioritizedMlpReplayBuffer
    from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy
    from offpolicy_amd.algorithms.maddpg.maddpg import MADDPG
    from bench import ddpg_transitions, DDPG_KEYS
    dims = DIMS["simple_spread"]
    args = default_args(use_per=True)
    pinfo = policy_info_for(dims)
    torch.manual_seed(1)
    np.random.seed(1)
    policy = MADDPGPolicy({"args": args, "device": dev}, pinfo["policy_0"])
    trainer = MADDPG(args, dims.n_agents, {"policy_0": policy}, lambda x: "policy_0", device=dev)
    cap = 4096
    buf = PrioritizedMlpReplayBuffer(args.per_alpha, pinfo, {"policy_0": list(range(dims.n_agents))}, cap, True, True, device=dev, device_tree=True)
    rng = np.random.RandomState(0)

    def insert(n):
        tr = ddpg_transitions(rng, n, dims)
        buf.insert(n, *[{"policy_0": tr[k]} for k in DDPG_KEYS])
    insert(2048)
    mem0, out = None, []
    for st in range(steps):
        if st % 10 == 0:
            insert(32)
        sample = buf.sample(256, min(1.0, 0.4 + st / steps), "policy_0")
        info, prio, idxes = trainer.shared_train_policy_on_batch("policy_0", sample)
        buf.update_priorities(idxes, prio, "policy_0")
        policy.soft_target_updates()
        if st % 500 == 499:
            cl = float(info["critic_loss"])
            assert np.isfinite(cl) and np.isfinite(float(info["critic_grad_norm"])), (st, cl)
            out.append(cl)
            mem = torch.cuda.memory_allocated(dev)
            mem0 = mem if mem0 is None else mem0
            assert mem <= mem0 + 64 * 2 ** 20, "device memory grows: %d -> %d" % (mem0, mem)
    print("maddpg simple_spread + PER: %d steps, critic loss every 500: %s, memory %.1f MB" % (steps, " ".join("%.4f" % x for x in out), mem0 / 2 ** 20))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    d = torch.device("cuda:0")
    qmix(n, d)
    maddpg(n, d)
