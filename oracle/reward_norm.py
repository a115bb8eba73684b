"""CPU oracle for the replay buffers' reward normalisation. TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Restates   RecPolicyBuffer.sample_inds   offpolicy/utils/rec_buffer.py:209-222   (episodes)
           MlpPolicyBuffer.sample_inds   offpolicy/utils/mlp_buffer.py:229-233   (transitions)
in numpy, float64 accumulation. Pinned by tests/golden/reward_norm.npz (outputs of the real reference buffers)."""
import numpy as np


def episode_reward_stats(rewards, dones_env, filled):
    """rewards [T, cap, N, 1], dones_env [T, cap, 1] (reference's time-major rings). Mean / population std over the steps of
    the first `filled` episodes whose PREVIOUS step did not end the episode (step 0 always counts), all agents."""
    T = rewards.shape[0]
    d = dones_env[:, :filled]
    curr = np.concatenate([np.zeros((1,) + d.shape[1:], d.dtype), d[:T - 1]], axis=0)        # [T, filled, 1]
    keep = np.broadcast_to((curr != 1.0)[:, :, None, :], rewards[:, :filled].shape)
    vals = rewards[:, :filled][keep].astype(np.float64)
    return vals.mean(), vals.std()


def transition_reward_stats(rewards, filled):
    vals = np.asarray(rewards[:filled], dtype=np.float64)
    return vals.mean(), vals.std()


def normalize(r, mean, std):
    return ((r.astype(np.float64) - mean) / std).astype(np.float32)
