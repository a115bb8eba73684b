"""Synthetic episode generator for benchmarks, parity tests and golden fixtures.

Follows the measurement recipe in SURVEY.md section 8(d): every config in BASELINE.json says "replay filled
with synthetic episodes", so no SMAC / MPE simulator is needed. Shapes are the ones the reference's
`RecPolicyBuffer.insert` receives (offpolicy/utils/rec_buffer.py:146-190): time-major
`[T(+1), episodes, agents, dim]`. Draw order from one `np.random.RandomState` is fixed (obs, share_obs,
action indices, rewards, lengths, [availability]) so fixtures regenerate bit-identically.
"""
from collections import namedtuple

import numpy as np

EnvDims = namedtuple("EnvDims", "name n_agents act_dim obs_dim state_dim episode_length")

# SURVEY.md Appendix B (StarCraft2_Env.py:1217-1318 + smac_maps.py) and MPE simple_spread.
DIMS = {
    "tiny": EnvDims("tiny", 2, 5, 12, 10, 6),
    "3m": EnvDims("3m", 3, 9, 64, 48, 60),
    "3s5z": EnvDims("3s5z", 8, 14, 252, 216, 150),
    "MMM2": EnvDims("MMM2", 10, 18, 370, 322, 180),
    "simple_spread": EnvDims("simple_spread", 3, 5, 18, 54, 25),
}
# --use_global_all_local_state (what the reference's QMIX-SMAC launch script runs, scripts/train_smac_qmix.sh:17): the centralized
# state also carries every agent's local observation, S = state + N * obs (StarCraft2_Env.py:1314-1315)
for _m in ("3m", "3s5z", "MMM2"):
    _d = DIMS[_m]
    DIMS[_m + "_gall"] = EnvDims(_m + "_gall", _d.n_agents, _d.act_dim, _d.obs_dim, _d.state_dim + _d.n_agents * _d.obs_dim, _d.episode_length)


def synth_episodes(rng, num_episodes, dims, avail="ones", runner_padding=False):
    """Draw `num_episodes` synthetic episodes.

    :param rng: np.random.RandomState
    :param dims: EnvDims
    :param avail: "ones" (all actions available) or "bernoulli" (p=0.8, action 0 always available and the
                  action actually taken always available)
    :param runner_padding: if True, mimic the runner's conventions past the episode end
                  (offpolicy/runner/rnn/smac_runner.py:65-71,110-134): avail_acts=0, acts=0, rewards=0.
    :return: dict of float32 arrays in insert() layout, plus int64 `lengths[E]`.
    """
    T, E, N = dims.episode_length, num_episodes, dims.n_agents
    A, D, S = dims.act_dim, dims.obs_dim, dims.state_dim
    obs = rng.standard_normal((T + 1, E, N, D)).astype(np.float32)
    share_obs = rng.standard_normal((T + 1, E, N, S)).astype(np.float32)
    # same-share convention: every agent sees the same centralized observation (insert keeps agent 0)
    share_obs[:] = share_obs[:, :, :1]
    act_idx = rng.randint(0, A, size=(T, E, N))
    rewards = np.repeat(rng.standard_normal((T, E, 1, 1)).astype(np.float32), N, axis=2)
    lengths = rng.randint(T // 2, T + 1, size=(E,))
    t_idx = np.arange(T)[:, None]
    dones_env = (t_idx >= (lengths[None, :] - 1)).astype(np.float32)[:, :, None]        # [T, E, 1]
    dones = np.repeat(dones_env[:, :, None, :], N, axis=2)                               # [T, E, N, 1]
    if avail == "ones":
        avail_acts = np.ones((T + 1, E, N, A), dtype=np.float32)
    elif avail == "bernoulli":
        avail_acts = (rng.random_sample((T + 1, E, N, A)) < 0.8).astype(np.float32)
        avail_acts[..., 0] = 1.0
        np.put_along_axis(avail_acts[:T], act_idx[..., None], 1.0, axis=-1)
    else:
        raise ValueError("unknown avail mode %r" % (avail,))
    acts = np.eye(A, dtype=np.float32)[act_idx]                                          # [T, E, N, A]
    if runner_padding:
        alive_t = (t_idx < lengths[None, :])                                             # [T, E]
        acts = acts * alive_t[:, :, None, None]
        rewards = rewards * alive_t[:, :, None, None]
        alive_t1 = (np.arange(T + 1)[:, None] <= lengths[None, :])
        avail_acts = avail_acts * alive_t1[:, :, None, None]
    return dict(obs=obs, share_obs=share_obs, acts=acts.astype(np.float32), rewards=rewards,
                dones=dones.astype(np.float32), dones_env=dones_env, avail_acts=avail_acts.astype(np.float32),
                lengths=lengths.astype(np.int64))


def policy_info_for(dims, continuous=False, multi_discrete=None):
    """`policy_info` dict in the SMAC style the reference builds at train_smac.py:131-137: list spaces. `continuous`: a Box action
    space of act_dim components in [-1, 1] instead of Discrete(act_dim); `multi_discrete`: the sizes of the sub-actions of a
    MultiDiscrete space (their sum = act_dim)."""
    from .spaces import Discrete, Box, MultiDiscrete
    act = Box(low=-np.ones(dims.act_dim, np.float32), high=np.ones(dims.act_dim, np.float32)) if continuous else Discrete(dims.act_dim)
    if multi_discrete is not None:
        assert int(np.sum(multi_discrete)) == dims.act_dim
        act = MultiDiscrete([[0, int(k) - 1] for k in multi_discrete])
    return {"policy_0": {"cent_obs_dim": dims.state_dim,
                         "cent_act_dim": dims.act_dim * dims.n_agents,
                         "obs_space": [dims.obs_dim],
                         "share_obs_space": [dims.state_dim],
                         "act_space": act}}


def as_policy_dicts(ep, p_id="policy_0"):
    """Wrap arrays the way runners hand them to `buffer.insert` (dict keyed by policy id)."""
    keys = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")
    return {k: {p_id: ep[k]} for k in keys}


def synth_fill_device(pbuf, n_episodes, dims, seed, avail="bernoulli", chunk=256):
    """Fill a device-resident RecPolicyBuffer (utils/rec_buffer.py) with `n_episodes` synthetic episodes generated ON THE DEVICE, in place
    in the store's episode-major rings: the same distributions as `synth_episodes` (obs / share_obs ~ N(0,1), uniform one-hot actions,
    rewards ~ N(0,1) shared by the agents, lengths uniform in [T/2, T] with dones_env = 1 from step L-1 on, Bernoulli(0.8) availability
    with action 0 always available), from torch's device generator seeded with `seed` -- every rank of a multi-GPU run gets the SAME
    store without pushing 7.5 GB of host-generated numbers through PCIe per rank (bench.py --gpus N). The ring
    counters advance exactly as `n_episodes` inserts would. Not a counterpart of anything in the reference: benchmark plumbing."""
    import torch
    T, N, A = dims.episode_length, dims.n_agents, dims.act_dim
    gen = torch.Generator(device=pbuf.device)
    gen.manual_seed(int(seed))
    done = 0
    while done < n_episodes:
        n = min(chunk, n_episodes - done)
        slots = torch.as_tensor(np.asarray(pbuf._ring.next_slots(n), dtype=np.int64), device=pbuf.device)
        pbuf.obs[slots] = torch.randn((n,) + tuple(pbuf.obs.shape[1:]), generator=gen, device=pbuf.device)
        pbuf.share_obs[slots] = torch.randn((n,) + tuple(pbuf.share_obs.shape[1:]), generator=gen, device=pbuf.device)
        idx = torch.randint(0, A, (n, T, N, 1), generator=gen, device=pbuf.device)
        pbuf.acts[slots] = torch.zeros((n, T, N, A), device=pbuf.device).scatter_(3, idx, 1.0)
        pbuf.rewards[slots] = torch.randn((n, T, 1, 1), generator=gen, device=pbuf.device).expand(n, T, N, 1)
        L = torch.randint(T // 2, T + 1, (n, 1), generator=gen, device=pbuf.device)
        de = (torch.arange(T, device=pbuf.device)[None, :] >= (L - 1)).to(torch.float32)          # [n, T]
        pbuf.dones_env[slots] = de[:, :, None]
        pbuf.dones[slots] = de[:, :, None, None].expand(n, T, N, 1)
        if pbuf.avail_acts is not None:
            if avail == "bernoulli":
                av = (torch.rand((n, T + 1, N, A), generator=gen, device=pbuf.device) < 0.8).to(torch.float32)
                av[..., 0] = 1.0
            else:
                av = torch.ones((n, T + 1, N, A), device=pbuf.device)
            pbuf.avail_acts[slots] = av
        done += n
    pbuf._stats_dirty = True
    pbuf._insert_gen += 1
    if getattr(pbuf, "_filled_dev", None) is not None:
        pbuf._filled_device()
