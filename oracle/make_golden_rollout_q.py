"""Rollout-side fixtures of the Q-LEARNING policies: what the REFERENCE's QMixPolicy / M_QMixPolicy return from get_actions /
get_random_actions for Discrete and MultiDiscrete action spaces (tests/golden/rollout_actions_q.npz).

TEST INFRASTRUCTURE ONLY. Run in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_rollout_q.py

Behind the q network, get_actions is host logic the engine mirrors in Python: availability masking, the greedy choice, and -- exploring --
one np.random.rand(batch) and one Categorical draw per call (Discrete) or per sub-action head (MultiDiscrete), in that order
(QMixPolicy.py:102-174, mQMixPolicy.py:62-113). Each record holds the reference network's q values for the call, both generators' states
before it, the flags and what the reference returned; tests/test_rollout_actions.py replays the engine's post-processing on the recorded q."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference, reference_args  # noqa: E402

load_reference()
from gym.spaces import Discrete  # noqa: E402
from offpolicy.utils.util import MultiDiscrete  # noqa: E402
from offpolicy.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy  # noqa: E402
from offpolicy.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "rollout_actions_q.npz")
STORE, NAMES = {}, []


def rng_state(pre):
    st = np.random.get_state()
    STORE[pre + "np_keys"], STORE[pre + "np_pos"] = st[1].copy(), np.array([st[2], st[3]], dtype=np.int64)
    STORE[pre + "np_gauss"], STORE[pre + "torch"] = np.array([st[4]]), torch.get_rng_state().numpy().copy()


def to_np(x):
    return x.detach().cpu().numpy().copy() if torch.is_tensor(x) else np.array(x, copy=True)


def main():
    args = reference_args(["--epsilon_anneal_time", "50"])
    dev = torch.device("cpu")
    D, S, B = 6, 9, 7
    for fam, Pol in (("rnn", QMixPolicy), ("mlp", M_QMixPolicy)):
        for kind, (space, width, heads) in {"discrete": (Discrete(5), 5, None), "md": (MultiDiscrete([[0, 2], [0, 3]]), 7, [3, 4])}.items():
            torch.manual_seed(11)
            np.random.seed(11)
            pinfo = {"cent_obs_dim": S, "cent_act_dim": 2 * width, "obs_space": [D], "share_obs_space": [S], "act_space": space}
            pol = Pol({"args": args, "device": dev}, pinfo)
            with torch.no_grad():      # away from the gain-0.01 heads: distinct greedy choices
                for p in pol.q_network.parameters():
                    p.add_(0.3 * torch.randn_like(p))
            rng = np.random.RandomState(5)
            calls = [("random", {}), ("act", dict(explore=True, t_env=3)), ("act", dict(explore=True, t_env=400)), ("act", dict(explore=False))]
            for ci, (what, kw) in enumerate(calls):
                name = "%s_%s/%d" % (fam, kind, ci)
                pre = name + "/"
                NAMES.append(name)
                obs = rng.standard_normal((B, D)).astype(np.float32)
                avail = None
                if kind == "discrete" and ci % 2 == 1:
                    avail = (rng.random_sample((B, width)) < 0.7).astype(np.float32)
                    avail[:, 0] = 1
                    STORE[pre + "avail"] = avail
                STORE[pre + "flags"] = np.array([what == "random", bool(kw.get("explore")), kw.get("t_env", -1)], dtype=np.int64)
                with torch.no_grad():
                    if fam == "rnn":
                        h0 = rng.standard_normal((B, 64)).astype(np.float32) * 0.3
                        q, _ = pol.get_q_values(torch.as_tensor(obs), np.zeros((B, width), np.float32), torch.as_tensor(h0))
                    else:
                        q = pol.get_q_values(torch.as_tensor(obs))
                    STORE[pre + "q"] = to_np(torch.cat(list(q), dim=-1) if isinstance(q, (list, tuple)) else q)
                    rng_state(pre)
                    if what == "random":
                        res, gq = pol.get_random_actions(obs, avail), None
                    elif fam == "rnn":
                        res, _, gq = pol.get_actions(torch.as_tensor(obs), np.zeros((B, width), np.float32), torch.as_tensor(h0), avail, **kw)
                    else:
                        res, gq = pol.get_actions(torch.as_tensor(obs), avail, **kw)
                STORE[pre + "actions"] = to_np(res)
                if gq is not None:
                    STORE[pre + "greedy_q"] = to_np(gq)
            STORE["%s_%s/heads" % (fam, kind)] = np.array(heads if heads else [], dtype=np.int64)
            STORE["%s_%s/width" % (fam, kind)] = np.array([width], dtype=np.int64)
    STORE["names"] = np.array(NAMES)
    STORE["hp"] = np.array([args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time])
    np.savez_compressed(OUT, **STORE)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(NAMES), "calls")


if __name__ == "__main__":
    main()
