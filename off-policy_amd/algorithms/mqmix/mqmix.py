"""MLP QMIX / VDN trainer on single transitions (mirror of offpolicy/algorithms/mqmix/mqmix.py:10-251, `M_QMix`).

Runs on the same HIP pipeline as the recurrent trainer with `ope_qmix_cfg.mlp = 1`: a transition is a one-step episode
whose time axis is (current, next), the GRU stage is skipped and the q head reads the trunk output directly. With T = 1
the recurrent formulas reduce exactly to the MLP ones: mask = 1, loss = mean over the batch (mqmix.py:188-205),
priorities = |error| + eps (mqmix.py:198).
"""
import torch

from ..qmix.qmix import QMix


def _stack_pair(cur, nxt, device):
    """[2, ...] contiguous stack of a (current, next) pair. When both are the two time slices of one gathered buffer
    (our MlpReplayBuffer), that buffer is returned as is -- no copy."""
    if torch.is_tensor(cur) and torch.is_tensor(nxt) and cur._base is not None and cur._base is nxt._base:
        base = cur._base
        if base.is_contiguous() and base.shape[0] == 2 and base.shape[1:].numel() == cur.numel() \
                and cur.data_ptr() == base.data_ptr() and nxt.data_ptr() == base.data_ptr() + 4 * cur.numel():
            return base
    c = torch.as_tensor(cur, dtype=torch.float32).to(device)
    n = torch.as_tensor(nxt, dtype=torch.float32).to(device)
    return torch.stack((c, n)).contiguous()


class M_QMix(QMix):
    _mlp = True

    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=None, vdn=False):
        super(M_QMix, self).__init__(args, num_agents, policies, policy_mapping_fn,
                                     device if device is not None else torch.device("cuda:0"), episode_length=1, vdn=vdn)

    def train_policy_on_batch(self, batch, use_same_share_obs=True):
        """See mqmix.py:68-218. `batch` is the 13-tuple of MlpReplayBuffer.sample()."""
        (obs_b, cent_b, act_b, rew_b, nobs_b, cent_nobs_b, dones_b, dones_env_b, valid_b, avail_b, navail_b,
         importance_weights, idxes) = batch
        pid = self.policy_ids[0]
        dev = self.device
        f = lambda x: torch.as_tensor(x, dtype=torch.float32).to(dev)
        if self.multi:       # several policies under one mixer (mqmix.py:95-178)
            parts = {}
            for q in self.policy_ids:
                avail = None
                if navail_b is not None and navail_b[q] is not None:
                    cur = avail_b[q] if (avail_b is not None and avail_b[q] is not None) else torch.ones_like(f(navail_b[q]))
                    avail = _stack_pair(cur, navail_b[q], dev)
                parts[q] = (_stack_pair(obs_b[q], nobs_b[q], dev), f(act_b[q])[None].contiguous(), avail)
            cent, ncent = (cent_b[pid], cent_nobs_b[pid]) if use_same_share_obs else (cent_b[pid][0], cent_nobs_b[pid][0])
            rew = f(rew_b[self.policy_ids[-1]])[None, :1].contiguous()      # mqmix.py:100,181: the LAST policy's agent 0
            return self._train_multi(parts, _stack_pair(cent, ncent, dev), rew, f(dones_env_b[pid])[None].contiguous(), importance_weights, idxes)
        obs = _stack_pair(obs_b[pid], nobs_b[pid], dev)                      # [2, N, B, D]
        # the mixer's state: the shared centralized observation, or agent 0's when every agent has its own (mqmix.py:78-84)
        cent, ncent = (cent_b[pid], cent_nobs_b[pid]) if use_same_share_obs else (cent_b[pid][0], cent_nobs_b[pid][0])
        share = _stack_pair(cent, ncent, dev)                                # [2, B, S]
        acts = f(act_b[pid])[None].contiguous()                              # [1, N, B, A]
        rew = f(rew_b[pid])[None].contiguous()                               # [1, N, B, 1]
        dones_env = f(dones_env_b[pid])[None].contiguous()                   # [1, B, 1]
        avail = None
        if navail_b is not None and navail_b[pid] is not None:
            cur = avail_b[pid] if (avail_b is not None and avail_b[pid] is not None) else torch.ones_like(f(navail_b[pid]))
            avail = _stack_pair(cur, navail_b[pid], dev)                     # only the NEXT slice is read (mqmix.py:158-160)
        if self._md_heads is not None:      # MultiDiscrete: one kernel-level agent per (agent, sub-action) (QMix._md_expand)
            obs, acts, rew, avail = self._md_expand(obs, acts, rew, avail)
        return self._train_on_device_batch(obs, share, acts, rew, dones_env, avail, importance_weights, idxes)
