"""Pin oracle/reward_norm.py against the REAL reference buffers' output (tests/golden/reward_norm.npz)."""
import numpy as np

from conftest import load_golden
from golden_util import EP_KEYS
from oracle import reward_norm as RN


def _ring(g, T, N, cap=8):
    """Replay the two inserts into the reference's time-major rings."""
    rewards = np.zeros((T, cap, N, 1), np.float32)
    dones_env = np.ones((T, cap, 1), np.float32)
    states = {}
    for tag in ("a", "b"):
        idx = g["rec_%s/idx_range" % tag]
        rewards[:, idx] = g["rec_%s/ep/rewards" % tag]
        dones_env[:, idx] = g["rec_%s/ep/dones_env" % tag]
        states[tag] = (rewards.copy(), dones_env.copy())
    return states


def test_episode_reward_normalisation_matches_reference():
    g = load_golden("reward_norm")
    N, _, _, _, T = [int(x) for x in g["rec_dims"]]
    states = _ring(g, T, N)
    for tag in ("a", "b"):
        rewards, dones_env = states[tag]
        filled = int(g["rec_%s/filled" % tag])
        mean, std = RN.episode_reward_stats(rewards, dones_env, filled)
        got = RN.normalize(rewards[:, g["rec_%s/inds" % tag]], mean, std).transpose(2, 0, 1, 3)     # _cast: [N, T, B, 1]
        np.testing.assert_allclose(got, g["rec_%s/rewards" % tag], rtol=2e-5, atol=2e-6)
    assert int(g["rec_b/filled"]) == 8 and int(g["rec_a/filled"]) == 5


def test_transition_reward_normalisation_matches_reference():
    g = load_golden("reward_norm")
    r = np.zeros((16,) + g["mlp/tr/rewards"].shape[1:], np.float32)
    r[g["mlp/idx_range"]] = g["mlp/tr/rewards"]
    mean, std = RN.transition_reward_stats(r, len(g["mlp/idx_range"]))
    got = RN.normalize(r[g["mlp/inds"]], mean, std).transpose(1, 0, 2)
    np.testing.assert_allclose(got, g["mlp/rewards"], rtol=2e-5, atol=2e-6)
