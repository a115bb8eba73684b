"""Rollout-side fixtures: what the REFERENCE's MADDPG-family policies return from get_actions / get_random_actions for every action-space
kind (Discrete, MultiDiscrete, Box), MLP and recurrent (tests/golden/rollout_actions.npz).

TEST INFRASTRUCTURE ONLY. Run in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_rollout.py

The part of get_actions that is NOT the actor network -- exploration noise, epsilon-greedy mixing, the order in which the numpy and torch
generators are consumed (MADDPGPolicy.py:63-119, rMADDPGPolicy.py:62-133) -- is host logic the engine mirrors in Python. Each record holds
the actor's output for the call (taken from the reference's own actor on the same input), both generators' states before the call, the
flags, and what the reference returned; tests/test_rollout_actions.py replays the engine's post-processing on the recorded actor output.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference, reference_args  # noqa: E402

load_reference()
from gym.spaces import Discrete, Box  # noqa: E402
from offpolicy.utils.util import MultiDiscrete  # noqa: E402
from offpolicy.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy  # noqa: E402
from offpolicy.algorithms.matd3.algorithm.MATD3Policy import MATD3Policy  # noqa: E402
from offpolicy.algorithms.r_maddpg.algorithm.rMADDPGPolicy import R_MADDPGPolicy  # noqa: E402
from offpolicy.algorithms.r_matd3.algorithm.rMATD3Policy import R_MATD3Policy  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "rollout_actions.npz")
STORE, NAMES = {}, []


def rng_state(pre):
    st = np.random.get_state()
    STORE[pre + "np_keys"], STORE[pre + "np_pos"] = st[1].copy(), np.array([st[2], st[3]], dtype=np.int64)
    STORE[pre + "np_gauss"], STORE[pre + "torch"] = np.array([st[4]]), torch.get_rng_state().numpy().copy()


def to_np(x):
    return x.detach().cpu().numpy().copy() if torch.is_tensor(x) else np.array(x, copy=True)


def spaces():
    return {"discrete": (Discrete(5), 5, None), "md": (MultiDiscrete([[0, 2], [0, 3]]), 7, [3, 4]),
            "cont": (Box(low=-np.ones(3, np.float32), high=np.ones(3, np.float32)), 3, None)}


def main():
    args = reference_args(["--epsilon_anneal_time", "50", "--act_noise_std", "0.15"])
    dev = torch.device("cpu")
    D, S, B = 6, 9, 7
    for fam, td3 in (("mlp", False), ("mlp", True), ("rnn", False), ("rnn", True)):
        for kind, (space, width, heads) in spaces().items():
            torch.manual_seed(11)
            np.random.seed(11)
            pinfo = {"cent_obs_dim": S, "cent_act_dim": 2 * width, "obs_space": [D], "share_obs_space": [S], "act_space": space}
            Pol = {("mlp", False): MADDPGPolicy, ("mlp", True): MATD3Policy, ("rnn", False): R_MADDPGPolicy, ("rnn", True): R_MATD3Policy}[(fam, td3)]
            pol = Pol({"args": args, "device": dev}, pinfo)
            rng = np.random.RandomState(5)
            calls = [("random", {}), ("act", dict(explore=True, t_env=3)), ("act", dict(explore=True, t_env=400)), ("act", dict(explore=False)),
                     ("act", dict(use_target=True)), ("act", dict(use_gumbel=True))]
            if kind == "cont":
                calls = [c for c in calls if not c[1].get("use_gumbel")]
            for ci, (what, kw) in enumerate(calls):
                name = "%s_%s_%s/%d" % (fam, "td3" if td3 else "ddpg", kind, ci)
                pre = name + "/"
                NAMES.append(name)
                obs = rng.standard_normal((B, D)).astype(np.float32)
                avail = None
                if kind == "discrete" and ci % 2 == 1:
                    avail = (rng.random_sample((B, width)) < 0.7).astype(np.float32)
                    avail[:, 0] = 1
                    STORE[pre + "avail"] = avail
                STORE[pre + "obs"] = obs
                STORE[pre + "flags"] = np.array([what == "random", bool(kw.get("explore")), bool(kw.get("use_target")), bool(kw.get("use_gumbel")),
                                                 kw.get("t_env", -1)], dtype=np.int64)
                with torch.no_grad():
                    if fam == "mlp":
                        net = pol.target_actor if kw.get("use_target") else pol.actor
                        out = net(obs)
                    else:
                        h0 = rng.standard_normal((B, 64)).astype(np.float32) * 0.3
                        STORE[pre + "h0"] = h0
                        net = pol.target_actor if kw.get("use_target") else pol.actor
                        out, _ = net(obs, np.zeros((B, width), np.float32), h0)
                    STORE[pre + "actor_out"] = to_np(torch.cat(list(out), dim=-1) if isinstance(out, (list, tuple)) else out)
                    rng_state(pre)
                    if what == "random":
                        res = pol.get_random_actions(obs, avail)
                        eps = None
                    elif fam == "mlp":
                        res, eps = pol.get_actions(obs, avail, **kw)
                    else:
                        res, _, eps = pol.get_actions(obs, np.zeros((B, width), np.float32), h0, avail, **kw)
                STORE[pre + "actions"] = to_np(res)
                STORE[pre + "eps"] = np.array([-1.0 if eps is None else float(eps)])
            STORE["%s_%s_%s/heads" % (fam, "td3" if td3 else "ddpg", kind)] = np.array(heads if heads else [], dtype=np.int64)
            STORE["%s_%s_%s/width" % (fam, "td3" if td3 else "ddpg", kind)] = np.array([width], dtype=np.int64)
    STORE["names"] = np.array(NAMES)
    STORE["hp"] = np.array([args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time, args.act_noise_std, args.target_action_noise_std])
    np.savez_compressed(OUT, **STORE)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(NAMES), "calls")


if __name__ == "__main__":
    main()
