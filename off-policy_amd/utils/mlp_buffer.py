"""Transition replay for MLP policies, resident in HBM.

Drop-in mirror of offpolicy/utils/mlp_buffer.py (`MlpReplayBuffer` 10-95, `MlpPolicyBuffer` 98-257,
`PrioritizedMlpReplayBuffer` 260-340): same constructors, 12-argument `insert`, 13-tuple `sample`.

A transition (obs, next_obs, ...) is stored as a ONE-STEP episode of the episode store (`RecPolicyBuffer` with
episode_length = 1): obs/share_obs/avail_acts have T+1 = 2 time entries = (current, next), everything else one. So the
same HIP insert/gather kernels serve both buffer families, and the gathered batch `[2][N][B][dim]` is already the
`[obs; next_obs]` stack the MLP trainers' kernels consume. `valid_transition` rides in a second, one-field store.
Outputs are CUDA tensors shaped like the reference's arrays ([N, B, dim] / [B, dim]); PER `insert` primes every new
slot (SURVEY.md A-3 fix).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .rec_buffer import RecPolicyBuffer, _shard
from .segment_tree import SumSegmentTree, MinSegmentTree


class MlpPolicyBuffer(object):
    def __init__(self, buffer_size, num_agents, obs_space, share_obs_space, act_space, use_same_share_obs, use_avail_acts,
                 use_reward_normalization=False, device=None):
        # transitions are one-step episodes of the device store; reward normalisation = plain mean/std (mlp_buffer.py:229-233)
        self._ep = RecPolicyBuffer(buffer_size, 1, num_agents, obs_space, share_obs_space, act_space, use_same_share_obs,
                                   use_avail_acts, use_reward_normalization, device=device, _reward_mask=False)
        self.buffer_size, self.num_agents = int(buffer_size), int(num_agents)
        self.use_same_share_obs, self.use_avail_acts = use_same_share_obs, use_avail_acts
        self.use_reward_normalization = use_reward_normalization
        self.device = self._ep.device
        # valid_transition defaults to zeros (mlp_buffer.py:155)
        self.valid_transition = torch.zeros((self.buffer_size, 1, self.num_agents, 1), dtype=torch.float32, device=self.device)

    @property
    def filled_i(self):
        return self._ep.filled_i

    @property
    def current_i(self):
        return self._ep.current_i

    @property
    def dims(self):
        return self._ep.dims

    def __len__(self):
        return self._ep.filled_i

    def _only_dones(self, t):
        f = _lib.Fields()
        f.dones = _lib.ptr(t).value
        return f

    def insert(self, num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env,
               valid_transition, avail_acts=None, next_avail_acts=None):
        obs = np.asarray(obs)
        assert obs.shape[0] == num_insert_steps, ("different size!")
        share_obs, next_share_obs = np.asarray(share_obs), np.asarray(next_share_obs)
        if self.use_same_share_obs and share_obs.ndim == 3:       # per-agent copies of the shared observation (mlp_buffer.py:176-178)
            share_obs, next_share_obs = share_obs[:, 0], next_share_obs[:, 0]
        av = np.stack((np.asarray(avail_acts), np.asarray(next_avail_acts))) if self.use_avail_acts else None
        idx_range = self._ep.insert(num_insert_steps, np.stack((obs, np.asarray(next_obs))),
                                    np.stack((share_obs, next_share_obs)), np.asarray(acts)[None], np.asarray(rewards)[None],
                                    np.asarray(dones)[None], np.asarray(dones_env)[None], av)
        n = int(num_insert_steps)
        staged = torch.from_numpy(np.ascontiguousarray(np.asarray(valid_transition, dtype=np.float32))[None]).to(self.device)
        slots = torch.from_numpy(np.asarray(idx_range, dtype=np.int64)).to(self.device)
        sf, df = self._only_dones(staged), self._only_dones(self.valid_transition)
        _lib.check(_lib.lib.ope_store_insert(C.byref(self._ep.dims), self.buffer_size, C.byref(df), C.byref(sf),
                                             _lib.ptr(slots), n, _lib.ptr(self._ep._bad_index), _lib.current_stream()), "ope_store_insert")
        self._keep = (staged, slots)
        return idx_range

    def sample_device(self, batch_size, seed, counter=None):
        """The 11-tuple of sample_inds for `batch_size` transitions drawn uniformly on the device (RecPolicyBuffer.sample_device),
        and the drawn indices."""
        inds = torch.empty(int(batch_size), dtype=torch.int64, device=self.device)
        return self.sample_inds(inds, _sampler=(int(seed), counter, self._ep._filled_device())), inds

    def sample_inds(self, sample_inds, timing_events=None, _sampler=None):
        """11-tuple of mlp_buffer.py:213-257: obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env,
        valid_transition, avail_acts, next_avail_acts (CUDA tensors, reference shapes)."""
        if torch.is_tensor(sample_inds):
            inds, B = sample_inds, int(sample_inds.shape[0])
        else:
            inds = np.asarray(sample_inds, dtype=np.int64)
            B = int(inds.shape[0])
        # the episode gather returns [N, T(+1), B, dim] views of [T(+1), N, B, dim] memory; T = 1 here. valid_transition rides in
        # the same launch as the 8th field.
        valid = torch.empty((1, self.num_agents, B, 1), dtype=torch.float32, device=self.device)
        obs, share, acts, rew, dones, dones_env, avail = self._ep.sample_inds(inds, timing_events=timing_events,
                                                                             extra=(self.valid_transition, valid), _sampler=_sampler)
        if not self.use_same_share_obs:   # per-agent centralized observations: [N, B, S] like the reference's _cast
            share = (share[:, 0], share[:, 1])
        return (obs[:, 0], share[0], acts[:, 0], rew[:, 0], obs[:, 1], share[1], dones[:, 0], dones_env[0], valid[0],
                avail[:, 0] if avail is not None else None, avail[:, 1] if avail is not None else None)


class MlpReplayBuffer(object):
    def __init__(self, policy_info, policy_agents, buffer_size, use_same_share_obs, use_avail_acts,
                 use_reward_normalization=False, device=None):
        self.policy_info = policy_info
        self.policy_buffers = {p_id: MlpPolicyBuffer(buffer_size, len(policy_agents[p_id]),
                                                     self.policy_info[p_id]['obs_space'],
                                                     self.policy_info[p_id]['share_obs_space'],
                                                     self.policy_info[p_id]['act_space'],
                                                     use_same_share_obs, use_avail_acts, use_reward_normalization, device=device)
                               for p_id in self.policy_info.keys()}

    def __len__(self):
        return self.policy_buffers['policy_0'].filled_i

    def insert(self, num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env,
               valid_transition, avail_acts, next_avail_acts):
        idx_range = None
        for p_id in self.policy_info.keys():
            idx_range = self.policy_buffers[p_id].insert(num_insert_steps, np.array(obs[p_id]), np.array(share_obs[p_id]),
                                                         np.array(acts[p_id]), np.array(rewards[p_id]),
                                                         np.array(next_obs[p_id]), np.array(next_share_obs[p_id]),
                                                         np.array(dones[p_id]), np.array(dones_env[p_id]),
                                                         np.array(valid_transition[p_id]),
                                                         np.array(avail_acts[p_id]), np.array(next_avail_acts[p_id]))
        return idx_range

    def _gather(self, inds):
        out = tuple({} for _ in range(11))
        for p_id in self.policy_info.keys():
            for dst, val in zip(out, self.policy_buffers[p_id].sample_inds(inds)):
                dst[p_id] = val
        return out

    def sample(self, batch_size, shard=None):
        """mlp_buffer.py:76-96. `shard=(rank, world)`: see RecReplayBuffer.sample (same global draw on every rank, own share gathered)."""
        inds = np.random.choice(len(self), batch_size)
        if shard is not None:
            inds = _shard(inds, *shard)
        return self._gather(inds) + (None, None)


class PrioritizedMlpReplayBuffer(MlpReplayBuffer):
    def __init__(self, alpha, policy_info, policy_agents, buffer_size, use_same_share_obs, use_avail_acts,
                 use_reward_normalization=False, device=None, device_tree=False):
        super(PrioritizedMlpReplayBuffer, self).__init__(policy_info, policy_agents, buffer_size, use_same_share_obs,
                                                         use_avail_acts, use_reward_normalization, device=device)
        self.alpha = alpha
        self.device_tree = bool(device_tree)
        it_capacity = 1
        while it_capacity < buffer_size:
            it_capacity *= 2
        if self.device_tree:
            from .device_per import DevicePerTree
            dev = self.policy_buffers[next(iter(self.policy_info))].device
            self._dtrees = {p_id: DevicePerTree(it_capacity, alpha, dev) for p_id in self.policy_info.keys()}
        else:
            self._it_sums = {p_id: SumSegmentTree(it_capacity) for p_id in self.policy_info.keys()}
            self._it_mins = {p_id: MinSegmentTree(it_capacity) for p_id in self.policy_info.keys()}
        self.max_priorities = {p_id: 1.0 for p_id in self.policy_info.keys()}

    def insert(self, num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones, dones_env,
               valid_transition, avail_acts=None, next_avail_acts=None):
        idx_range = super().insert(num_insert_steps, obs, share_obs, acts, rewards, next_obs, next_share_obs, dones,
                                   dones_env, valid_transition, avail_acts, next_avail_acts)
        for p_id in self.policy_info.keys():       # every new slot (A-3 fix)
            if self.device_tree:
                self._dtrees[p_id].set_to_max(idx_range)
                continue
            self._it_sums[p_id][idx_range] = self.max_priorities[p_id] ** self.alpha
            self._it_mins[p_id][idx_range] = self.max_priorities[p_id] ** self.alpha
        return idx_range

    def _sample_proportional(self, batch_size, p_id=None):
        total = self._it_sums[p_id].sum(0, len(self) - 1)
        mass = np.random.random(size=batch_size) * total
        return self._it_sums[p_id].find_prefixsum_idx(mass)

    def sample(self, batch_size, beta=0, p_id=None, shard=None):
        """`shard=(rank, world)`: see PrioritizedRecReplayBuffer.sample (own share of episodes / weights, GLOBAL indices)."""
        assert len(self) > batch_size, "Not enough samples in the buffer!"
        assert beta > 0
        if self.device_tree:    # same host RNG draw as _sample_proportional; tree walk and weights on the device
            inds, weights = self._dtrees[p_id].sample(np.random.random(size=batch_size), len(self), beta)
            if shard is not None:
                return self._gather(_shard(inds, *shard).contiguous()) + (_shard(weights, *shard).contiguous(), inds)
            return self._gather(inds) + (weights, inds)
        batch_inds = self._sample_proportional(batch_size, p_id)
        p_min = self._it_mins[p_id].min() / self._it_sums[p_id].sum()
        max_weight = (p_min * len(self)) ** (-beta)
        p_sample = self._it_sums[p_id][batch_inds] / self._it_sums[p_id].sum()
        weights = (p_sample * len(self)) ** (-beta) / max_weight
        if shard is not None:
            return self._gather(_shard(batch_inds, *shard)) + (_shard(weights, *shard), batch_inds)
        return self._gather(batch_inds) + (weights, batch_inds)

    def sample_device(self, batch_size, beta_dev, p_id=None):
        """Device-tree prioritized sample with no host data (capturable in a HIP graph): masses from torch.rand, the filled count from
        the buffer's device counter, beta from `beta_dev` (float64 [1] device tensor). Returns the 11-tuple + (weights, indices)."""
        assert self.device_tree, "device-side prioritized sampling needs device_tree=True"
        pbuf = self.policy_buffers[p_id]
        mass = torch.rand(int(batch_size), dtype=torch.float64, device=pbuf.device)
        inds, weights = self._dtrees[p_id].sample_dev(mass, pbuf._ep._filled_device(), beta_dev)
        return tuple({p_id: x} for x in pbuf.sample_inds(inds)) + (weights, inds)

    def update_priorities(self, idxes, priorities, p_id=None):
        if self.device_tree:    # range checks of the host path would force a device sync; the kernels clamp nothing: callers pass
            self._dtrees[p_id].set(idxes, priorities)   # indices returned by sample()
            return
        priorities, idxes = np.asarray(priorities), np.asarray(idxes)
        assert len(idxes) == len(priorities)
        assert np.min(priorities) > 0
        assert np.min(idxes) >= 0
        assert np.max(idxes) < len(self)
        self._it_sums[p_id][idxes] = priorities ** self.alpha
        self._it_mins[p_id][idxes] = priorities ** self.alpha
        self.max_priorities[p_id] = max(self.max_priorities[p_id], np.max(priorities))
