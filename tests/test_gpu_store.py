"""-m gpu: replay store insert/gather through the C-ABI, bit-exact against the oracle / reference fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from golden_util import reference_store_from
from gpu_util import build_from_fixture

pytestmark = pytest.mark.gpu
FIELDS = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")


@pytest.mark.parametrize("name", ["qmix_tiny", "qmix_tiny_huber_per", "qmix_odd", "vdn_tiny"])
def test_gather_bit_exact_vs_reference_fixture(name):
    """insert (with ring wrap) + sample_inds == what the reference's RecPolicyBuffer returned, bit for bit."""
    g = load_golden(name)
    dims, buf, _, _ = build_from_fixture(g)
    got = buf.policy_buffers["policy_0"].sample_inds(g["inds"])
    for k, a in zip(FIELDS, got):
        ref = g["batch/" + k]
        assert tuple(a.shape) == ref.shape, k
        assert np.array_equal(a.cpu().numpy(), ref), k


@pytest.mark.parametrize("dims_name,B", [("3m", 32), ("3s5z", 32), ("MMM2", 8), ("simple_spread", 16)])
def test_gather_matches_oracle_at_scale(dims_name, B):
    """Full-size configs: HIP gather vs the numpy oracle on the same seeded store (bit-exact), repeated + wrapped indices."""
    from oracle.qmix_oracle import sample_inds
    from offpolicy_amd.utils.synth import DIMS, synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    dims = DIMS[dims_name]
    cap = 40
    rng = np.random.RandomState(5)
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, True, True,
                          device="cuda:0")
    T, N, A, D, S = dims.episode_length, dims.n_agents, dims.act_dim, dims.obs_dim, dims.state_dim
    st = dict(obs=np.zeros((T + 1, cap, N, D), np.float32), share_obs=np.zeros((T + 1, cap, S), np.float32),
              acts=np.zeros((T, cap, N, A), np.float32), avail_acts=np.ones((T + 1, cap, N, A), np.float32),
              rewards=np.zeros((T, cap, N, 1), np.float32), dones=np.ones((T, cap, N, 1), np.float32),
              dones_env=np.ones((T, cap, 1), np.float32))
    for n_ins in (25, 30):            # second insert wraps around
        ep = synth_episodes(rng, n_ins, dims, avail="bernoulli")
        d = as_policy_dicts(ep)
        idx = buf.insert(n_ins, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
        for k in FIELDS:
            st[k][:, idx] = ep[k][:, :, 0] if k == "share_obs" else ep[k]
    inds = rng.randint(0, len(buf), size=B)
    inds[1] = inds[0]
    got = buf.policy_buffers["policy_0"].sample_inds(inds)
    want = sample_inds(st, inds)
    for k, a, w in zip(FIELDS, got, want):
        assert np.array_equal(a.cpu().numpy(), w), k


def test_gather_untouched_slots_keep_reference_defaults():
    """Slots never written read back as the reference's initial values (zeros; ones for avail/dones/dones_env)."""
    from offpolicy_amd.utils.synth import DIMS, policy_info_for
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    dims = DIMS["tiny"]
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": [0, 1]}, 4, dims.episode_length, True, True, device="cuda:0")
    obs, share, acts, rew, dones, dones_env, avail = buf.policy_buffers["policy_0"].sample_inds(np.array([0, 3]))
    assert float(obs.abs().sum()) == 0 and float(acts.abs().sum()) == 0 and float(rew.abs().sum()) == 0
    assert bool((avail == 1).all()) and bool((dones == 1).all()) and bool((dones_env == 1).all())
