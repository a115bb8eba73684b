"""Golden fixture for the buffers' reward normalisation, generated with the REAL reference buffers on CPU.
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_reward_norm.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle.ref_import import load_reference  # noqa: E402

load_reference()
from gym.spaces import Discrete  # noqa: E402
from offpolicy.utils.rec_buffer import RecReplayBuffer  # noqa: E402
from offpolicy.utils.mlp_buffer import MlpReplayBuffer  # noqa: E402
from make_golden_mlp import synth_transitions, T_KEYS  # noqa: E402

from offpolicy_amd.utils.synth import EnvDims, synth_episodes, as_policy_dicts  # noqa: E402

KEYS = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")


def main():
    out = {}
    dims = EnvDims("rn", 3, 4, 6, 5, 7)
    pinfo = {"policy_0": {"cent_obs_dim": dims.state_dim, "cent_act_dim": dims.act_dim * dims.n_agents, "obs_space": [dims.obs_dim],
                          "share_obs_space": [dims.state_dim], "act_space": Discrete(dims.act_dim)}}
    agents = {"policy_0": list(range(dims.n_agents))}
    rng = np.random.RandomState(11)
    # episodes: partially filled ring (5 of 8), then wrapped (5 + 6 into 8)
    buf = RecReplayBuffer(pinfo, agents, 8, dims.episode_length, True, True, True)
    for tag, n in (("a", 5), ("b", 6)):
        ep = synth_episodes(rng, n, dims, avail="bernoulli", runner_padding=True)
        ep["rewards"] = (ep["rewards"] * 3.0 + 1.5).astype(np.float32) * (ep["rewards"] != 0)
        d = as_policy_dicts(ep)
        out["rec_%s/idx_range" % tag] = np.asarray(buf.insert(n, *[d[k] for k in KEYS]))
        for k in KEYS:
            out["rec_%s/ep/%s" % (tag, k)] = ep[k]
        inds = np.array([0, 3, 3, 1, 4], dtype=np.int64)
        s = buf.policy_buffers["policy_0"].sample_inds(inds)
        out["rec_%s/inds" % tag] = inds
        out["rec_%s/rewards" % tag] = np.ascontiguousarray(s[3]).astype(np.float32)
        out["rec_%s/filled" % tag] = np.int64(buf.policy_buffers["policy_0"].filled_i)
    out["rec_dims"] = np.array([dims.n_agents, dims.act_dim, dims.obs_dim, dims.state_dim, dims.episode_length], dtype=np.int64)
    # transitions
    tdims = EnvDims("rnt", 2, 4, 10, 12, 1)
    tinfo = {"policy_0": {"cent_obs_dim": tdims.state_dim, "cent_act_dim": tdims.act_dim * tdims.n_agents, "obs_space": [tdims.obs_dim],
                          "share_obs_space": [tdims.state_dim], "act_space": Discrete(tdims.act_dim)}}
    mbuf = MlpReplayBuffer(tinfo, {"policy_0": list(range(tdims.n_agents))}, 16, True, True, True)
    tr = synth_transitions(rng, 11, tdims)
    tr["rewards"] = (tr["rewards"] * 2.0 - 0.7).astype(np.float32)
    out["mlp/idx_range"] = np.asarray(mbuf.insert(11, *[{"policy_0": tr[k]} for k in T_KEYS]))
    for k in T_KEYS:
        out["mlp/tr/" + k] = tr[k]
    inds = np.array([10, 0, 7, 7, 2], dtype=np.int64)
    s = mbuf.policy_buffers["policy_0"].sample_inds(inds)
    out["mlp/inds"] = inds
    out["mlp/rewards"] = np.ascontiguousarray(s[3]).astype(np.float32)
    out["mlp_dims"] = np.array([tdims.n_agents, tdims.act_dim, tdims.obs_dim, tdims.state_dim, 1], dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "reward_norm.npz")
    np.savez_compressed(path, **out)
    print("reward_norm.npz %.0f KB; rec rewards a mean %.4f std %.4f" % (os.path.getsize(path) / 1024.0, out["rec_a/rewards"].mean(), out["rec_a/rewards"].std()))


if __name__ == "__main__":
    main()
