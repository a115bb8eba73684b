"""Recurrent MADDPG / MATD3 trainer on the HIP engine.

Mirror of offpolicy/algorithms/r_maddpg/r_maddpg.py:9-331 (`R_MADDPG`, shared-observation path
`shared_train_policy_on_batch`) for discrete one-hot actions. Per update:

    ope_rddpg_critic_loss_and_grad -> ope_adam_step(critic)
    [every actor_update_interval-th update]  ope_rddpg_actor_loss_and_grad -> ope_adam_step(actor)

One shared policy ('policy_0' for all agents) or one policy per group of agents (share_policy = False, what
scripts/train_mpe_rmaddpg.sh runs): then every policy has its own actor / critic / buffer, an update of policy p first
collects the target actions of EVERY policy's target actor (ope_rddpg_target_actions, get_update_info r_maddpg.py:44-105)
into the joint next action, trains p's critic on it and on the joint buffer action of all agents, and in the actor update
replaces only p's agents' blocks (act_sequence_replace_ind_start). All policies must have the same act_dim.

The reference walks the target critic (critic update) and the live critic (actor update) through the episode with a
Python loop of 2*T network calls; the engine runs the buffer-sequence scan once and all "sideways" steps as one
row-parallel GRU-cell launch (csrc/ope_rddpg.hip). Gumbel noise is drawn on the CPU generator in the reference's
order and shapes: target noise [(T+1), N*B, A] first (only when target_noise is set), then actor noise [T, N*B, A].
Unlike the MLP trainer, `num_updates` IS incremented here (r_maddpg.py:330), so R_MATD3 really
delays its actor update.
"""
import ctypes as C

import numpy as np
import torch

from ...utils.rec_buffer import StoreObs

from ... import _lib
from ... import dist as opdist
from ..maddpg.algorithm.MADDPGPolicy import sample_gumbel_uniform


class R_MADDPG(object):
    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=None, episode_length=None, actor_update_interval=1):
        self.args = args
        if getattr(args, "use_popart", False):
            raise NotImplementedError("use_popart is not on the accelerated path")
        self.use_per, self.per_eps = args.use_per, args.per_eps
        self.use_huber_loss, self.huber_delta = args.use_huber_loss, args.huber_delta
        self.device = torch.device(device if device is not None else "cuda:0")
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.episode_length = args.episode_length if episode_length is None else episode_length
        self.num_agents, self.policies, self.policy_mapping_fn = num_agents, policies, policy_mapping_fn
        self.policy_ids = sorted(list(self.policies.keys()))
        self.policy_agents = {pid: sorted([a for a in range(num_agents) if policy_mapping_fn(a) == pid]) for pid in self.policies}
        self.multi_policy = len(self.policy_ids) > 1
        # joint-action order = policy order (get_update_info iterates self.policy_ids), each policy's agents in turn
        self.agent_offset, off = {}, 0
        for pid in self.policy_ids:
            self.agent_offset[pid] = off
            off += len(self.policy_agents[pid])
        assert off == num_agents, "every agent must be mapped to a policy"
        # policies of different action dimensions (simple_speaker_listener under scripts/train_mpe_rmaddpg.sh): the joint action is then
        # described in columns (ope_rddpg_cfg.joint_act_dim / joint_act_col / joint_acts) instead of equal agent blocks
        self.mixed_act_dims = len({self.policies[pid].output_dim for pid in self.policy_ids}) != 1
        self.joint_act_col, col = {}, 0
        for pid in self.policy_ids:
            self.joint_act_col[pid] = col
            col += len(self.policy_agents[pid]) * self.policies[pid].output_dim
        self.joint_act_dim = col
        self.actor_update_interval = actor_update_interval
        self.num_updates = {p_id: 0 for p_id in self.policy_ids}
        self.use_same_share_obs = args.use_same_share_obs
        # False: gumbel noise from torch's CPU generator in the reference's order (bit-comparable runs, ~2x(T*N*B*A) floats
        # drawn on the host and copied per update). True: same distribution drawn on the device (no host work).
        self.device_noise = False
        # (target noise [(T+1), N*B, A] or None, actor noise [T, N*B, A]) uniform draws to consume INSTEAD of drawing: lets a
        # sharded (data-parallel) run use the columns of one full-batch realisation; consumed by one update, then cleared
        self._noise_override = None
        self._ws, self._grads = {}, {}

    def _workspace(self, policy, cfg):
        # one workspace + gradient / Adam-scratch set PER POLICY: the buffers below are sized from this policy's actor and critic
        # (policies of a multi-policy trainer may differ in observation width while sharing batch size and agent counts)
        B = (cfg.batch, cfg.dims.n_agents, cfg.n_total_agents, cfg.joint_act_dim, id(policy)) if (cfg.n_total_agents or cfg.joint_act_dim) else cfg.batch
        if B not in self._ws:
            need = _lib.lib.ope_rddpg_workspace_bytes(C.byref(cfg))
            if need < 0:
                _lib.check(int(need), "ope_rddpg_workspace_bytes")
            ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib.ope_rddpg_workspace_init(C.byref(cfg), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                       "ope_rddpg_workspace_init")
            self._ws[B] = ws
            nmax = max(policy.critic.padded_numel, policy.actor.padded_numel)
            # (the critic's vector has room behind its tail for the ranks' per-episode priorities: dist.priority_slots)
            self._grads[B] = (torch.zeros(policy.critic.padded_numel + 4 + cfg.batch * opdist.world()[1], **self.tpdv), torch.zeros(policy.actor.padded_numel + 4, **self.tpdv),
                              torch.zeros(int(_lib.lib.ope_adam_scratch_floats(nmax)), **self.tpdv))
        return self._ws[B], self._grads[B]

    def workspace_view(self, batch, name, policy_id="policy_0"):
        cfg = self.policies[policy_id].rddpg_cfg(batch, self.episode_length)
        n = C.c_int64(0)
        off = _lib.lib.ope_rddpg_workspace_find(C.byref(cfg), name.encode(), C.byref(n))
        if off < 0:
            raise KeyError(name)
        return self._ws[batch][off:off + 4 * n.value].view(torch.float32)

    def train_policy_on_batch(self, update_policy_id, batch):
        if self.use_same_share_obs:
            return self.shared_train_policy_on_batch(update_policy_id, batch)
        return self.cent_train_policy_on_batch(update_policy_id, batch)

    def cent_train_policy_on_batch(self, update_policy_id, batch):
        """Every agent has its OWN centralized observation (use_same_share_obs = False): r_maddpg.py:333-564. `batch` = the 9-tuple of
        RecReplayBuffer.sample() with cent_obs [N, T+1, B, S]. Upstream this function fails on every input (the
        dispatcher passes one tuple, and lines 355-356 slice the agent axis where the time axis is meant); the semantics here are
        the ones the code plainly intends, pinned on outputs of the reference run with that reading (oracle/make_golden_cent.py,
        tests/golden/r*_cent_*.npz).
        The critic is trained on N*B episodes -- episode (i, b) = episode b seen through agent i's observation, joint actions,
        rewards and dones repeated (lines 361-379) -- and in the actor update copy i of episode b carries agent i's observation
        (line 544). Both are the shared-observation update on a batch laid out as N*B episodes, with the actor objective's copy
        `rep` of episode (i, b) counted only for rep == i (ope_rddpg_cfg.actor_row_weight): the same C-ABI calls, correct rather
        than tuned -- the actor pass evaluates N times the rows it needs."""
        obs_b, cent_b, act_b, rew_b, dones_b, dones_env_b, avail_b, importance_weights, idxes = batch
        pid = update_policy_id
        if self.multi_policy:
            raise NotImplementedError("cent_train_policy_on_batch with several policies is not on the accelerated path")
        if self.use_per:
            raise NotImplementedError("cent_train_policy_on_batch returns N*B priorities for B indices upstream (r_maddpg.py:447-449), "
                                      "which the buffer rejects: uniform replay only")
        if getattr(self.args, "use_value_active_masks", False):
            raise NotImplementedError("cent_train_policy_on_batch with use_value_active_masks: upstream weights the critic loss by the agents' "
                                      "active masks there (r_maddpg.py:418-498); the accelerated path takes the plain masked mean")
        policy = self.policies[pid]
        if policy.multidiscrete or not policy.discrete:
            raise NotImplementedError("cent_train_policy_on_batch with a continuous / multi-discrete action space is not on the accelerated path "
                                      "(no reference fixture pins it)")
        obs = self._to_device_layout(obs_b[pid], True)                      # [T+1, N, B, D]
        cent = self._to_device_layout(cent_b[pid], True)                    # [T+1, N, B, S]
        acts = self._to_device_layout(act_b[pid], True)
        rew = self._to_device_layout(rew_b[pid], True)
        dones = self._to_device_layout(dones_b[pid], True)
        dones_env = self._to_device_layout(dones_env_b[pid], False)         # [T, B, 1]
        avail = self._to_device_layout(avail_b[pid], True) if (avail_b is not None and avail_b[pid] is not None) else None
        T1, N, B, _ = obs.shape
        T, A = self.episode_length, policy.output_dim
        tile = lambda x: None if x is None else x.repeat(1, 1, N, 1).contiguous()     # [., N, N*B, .]: column (i, b) <- b
        share_v = cent.reshape(T1, N * B, cent.shape[-1]).contiguous()                 # column (i, b) = agent i's observation of b
        eye = torch.eye(N, **self.tpdv).repeat_interleave(B, dim=1).contiguous()       # [N, N*B]: copy a of episode (i, b) counts iff a == i
        # the reference draws its noise for the N*B real rows ([T(+1), N*B, A], target noise first): same draws, repeated per copy
        draw = (lambda shape: torch.rand(shape, **self.tpdv)) if self.device_noise else (lambda shape: sample_gumbel_uniform(shape).to(self.device))
        rep = lambda u, L: u.view(L, N, B, A).repeat(1, 1, N, 1).view(L, N * N * B, A).contiguous()
        u_t = rep(draw((T + 1, N * B, A)), T + 1) if policy.target_noise is not None else None
        update_actor = self.num_updates[pid] % self.actor_update_interval == 0
        u_a = rep(draw((T, N * B, A)), T) if update_actor else None
        self._noise_override = (u_t, u_a)
        self._actor_row_weight = eye
        try:
            return self._train_on_device_batch(policy, pid, tile(obs), share_v, tile(acts), tile(rew), tile(dones),
                                               dones_env.repeat(1, N, 1).contiguous(), tile(avail), None, idxes)
        finally:
            self._actor_row_weight = None

    def _adam(self, opt, n, flat, flat_tgt, grad, scratch, qden, skip=(0, 0)):
        opt.step_count += 1
        ac = _lib.AdamCfg()
        ac.lr, ac.beta1, ac.beta2, ac.eps = opt.lr, opt.betas[0], opt.betas[1], opt.eps
        ac.max_grad_norm, ac.weight_decay, ac.tau, ac.do_polyak = float(self.args.max_grad_norm), float(getattr(self.args, "weight_decay", 0.0)), 0.0, 0
        ac.skip_begin, ac.skip_end = skip          # the unused fc_h block: torch's Adam never touches grad-less tensors
        ac.step, ac.qtot_denominator, ac.tail_offset = opt.step_count, float(qden), int(n)
        stats = torch.empty(4, **self.tpdv)
        _lib.check(_lib.lib.ope_adam_step(C.byref(ac), int(n), _lib.ptr(flat), _lib.ptr(flat_tgt), _lib.ptr(opt.exp_avg),
                                          _lib.ptr(opt.exp_avg_sq), _lib.ptr(grad), _lib.ptr(scratch), _lib.ptr(stats),
                                          _lib.current_stream()), "ope_adam_step")
        return stats

    def _to_device_layout(self, x, agent_axis):
        """[N, T(+1), B, dim] (reference's per-agent stacking) -> the kernels' [T(+1), N, B, dim]; [T(+1), B, dim] as is."""
        if x is None:
            return None
        if isinstance(x, StoreObs):      # observations left in the store (RecPolicyBuffer.lazy_obs): gathered on the device, no host trip
            x = x.materialize()
        if torch.is_tensor(x):
            t = x.to(self.device, dtype=torch.float32)
            if agent_axis:
                t = t.permute(1, 0, 2, 3)
            return t if t.is_contiguous() else t.contiguous()
        a = np.asarray(x, dtype=np.float32)
        if agent_axis:
            a = a.transpose(1, 0, 2, 3)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def shared_train_policy_on_batch(self, update_policy_id, batch):
        """See r_maddpg.py:114-331. `batch` = 9-tuple of RecReplayBuffer.sample()."""
        obs_b, cent_b, act_b, rew_b, dones_b, dones_env_b, avail_b, importance_weights, idxes = batch
        pid = update_policy_id
        policy = self.policies[pid]
        obs = self._to_device_layout(obs_b[pid], True)
        share = self._to_device_layout(cent_b[pid], False)
        acts = self._to_device_layout(act_b[pid], True)
        rew = self._to_device_layout(rew_b[pid], True)
        dones = self._to_device_layout(dones_b[pid], True)
        dones_env = self._to_device_layout(dones_env_b[pid], False)
        avail = self._to_device_layout(avail_b[pid], True) if (avail_b is not None and avail_b[pid] is not None) else None
        others = None
        if self.multi_policy:       # every policy's observations (for its target actor) and buffer actions (joint action), policy order
            others = []
            for q in self.policy_ids:
                o_q = obs if q == pid else self._to_device_layout(obs_b[q], True)
                a_q = acts if q == pid else self._to_device_layout(act_b[q], True)
                v_q = avail if q == pid else (self._to_device_layout(avail_b[q], True) if (avail_b is not None and avail_b[q] is not None) else None)
                others.append((q, o_q, a_q, v_q))
        return self._train_on_device_batch(policy, pid, obs, share, acts, rew, dones, dones_env, avail, importance_weights, idxes, others)

    def _train_on_device_batch(self, policy, pid, obs, share, acts, rew, dones, dones_env, avail, importance_weights, idxes, others=None):
        T1, N, B, D = obs.shape
        T = self.episode_length
        assert T1 == T + 1 and N == len(self.policy_agents[pid]), "batch does not match the trainer's dimensions"
        A = policy.output_dim
        cfg = policy.rddpg_cfg(B, T)
        draw = (lambda shape: torch.rand(shape, **self.tpdv)) if self.device_noise else (lambda shape: sample_gumbel_uniform(shape).to(self.device))

        def noise_for(pol, L, rows, target):
            """What pol.get_actions draws for L steps of `rows` rows (rMADDPGPolicy.py:81-129), on the generator the reference uses: a
            multi-discrete policy one uniform block per sub-action in order (a gumbel_softmax per head), a continuous one gaussian noise on its
            TARGET actions only, a discrete one one uniform block."""
            if not pol.discrete:
                from ..maddpg.algorithm.MADDPGPolicy import gaussian_noise
                assert not self.device_noise, "continuous actions: host noise"
                return gaussian_noise((L, rows, pol.output_dim), float(pol.target_noise)).to(self.device) if (target and pol.target_noise is not None) else None
            if pol.multidiscrete:
                assert not self.device_noise, "multi-discrete actions: host noise"
                return torch.cat([sample_gumbel_uniform((L, rows, int(k))) for k in pol.act_dim], dim=-1).to(self.device).contiguous()
            return draw((L, rows, pol.output_dim))
        st = _lib.current_stream()
        joint_next = keep = None
        if others is not None:
            # joint target action: one ope_rddpg_target_actions per policy (its target actor over its agents' T+1 observations; noise
            # drawn per policy in policy order, as get_update_info does), scattered into [T][B][N_total * A]
            NT = self.num_agents
            mixed = self.mixed_act_dims
            joint_next = torch.empty(T, B, self.joint_act_dim if mixed else NT * A, **self.tpdv)
            keep = []
            for q, o_q, a_q, v_q in others:
                pol_q = self.policies[q]
                cq = pol_q.rddpg_cfg(B, T)
                cq.dims.n_agents, cq.n_total_agents, cq.agent_offset = o_q.shape[1], NT, self.agent_offset[q]
                if mixed:
                    cq.n_total_agents, cq.agent_offset, cq.joint_act_dim, cq.joint_act_col = 0, 0, self.joint_act_dim, self.joint_act_col[q]
                ws_q, _ = self._workspace(pol_q, cq)
                fq = _lib.Fields()
                fq.obs, fq.avail_acts = _lib.ptr(o_q).value, _lib.ptr(v_q).value
                u_q = noise_for(pol_q, T + 1, o_q.shape[1] * B, True) if pol_q.target_noise is not None else None
                _lib.check(_lib.lib.ope_rddpg_target_actions(C.byref(cq), C.byref(fq), _lib.ptr(pol_q.target_actor._flat), _lib.ptr(u_q),
                                                             _lib.ptr(ws_q), ws_q.numel(), _lib.ptr(joint_next), st), "ope_rddpg_target_actions")
                keep.append((u_q, o_q, v_q))
            if mixed:      # the buffer's joint action [T][B][sum of widths]: every agent's block side by side, policy order
                joint_acts = torch.cat([a_q.permute(0, 2, 1, 3).reshape(T, B, -1) for _, _, a_q, _ in others], dim=-1).contiguous()
                assert joint_acts.shape[-1] == self.joint_act_dim
                keep.append(joint_acts)
                cfg.dims.n_agents, cfg.joint_act_dim, cfg.joint_act_col = N, self.joint_act_dim, self.joint_act_col[pid]
                cfg.joint_acts = _lib.ptr(joint_acts).value
            else:
                acts = torch.cat([a_q for _, _, a_q, _ in others], dim=1).contiguous()      # [T][N_total][B][A]
                cfg.dims.n_agents, cfg.n_total_agents, cfg.agent_offset = N, NT, self.agent_offset[pid]
            cfg.joint_next_acts = _lib.ptr(joint_next).value
        roww = getattr(self, "_actor_row_weight", None)
        if roww is not None:
            cfg.actor_row_weight = _lib.ptr(roww).value
        ws, (gc, ga, scratch) = self._workspace(policy, cfg)
        f = _lib.Fields()
        f.obs, f.share_obs, f.acts, f.rewards = _lib.ptr(obs).value, _lib.ptr(share).value, _lib.ptr(acts).value, _lib.ptr(rew).value
        f.dones, f.dones_env, f.avail_acts = _lib.ptr(dones).value, _lib.ptr(dones_env).value, _lib.ptr(avail).value
        _, world_size = opdist.world()
        train_info = {}
        update_actor = self.num_updates[pid] % self.actor_update_interval == 0
        # ---- critic ----
        override, self._noise_override = self._noise_override, None
        if others is not None:
            assert override is None, "noise overrides are for the single-policy data-parallel path"
            u_t = None
        elif override is not None:
            u_t = None if override[0] is None else override[0].to(self.device, dtype=torch.float32).contiguous()
        else:
            # (continuous actions, rMADDPGPolicy.py:121-129: additive gaussian noise on the target action when the policy has a target noise
            #  -- R_MATD3 --, drawn on the CPU generator in the reference's order and shape; nothing is drawn for the actor update)
            u_t = noise_for(policy, T + 1, N * B, True) if policy.target_noise is not None else None
        if not policy.discrete:
            assert override is None, "continuous actions: one process"
        dev_prio = torch.is_tensor(importance_weights)     # device trees: weights in, priorities out stay in HBM
        w = None
        if self.use_per:
            w = (importance_weights.to(self.device, dtype=torch.float32).contiguous() if dev_prio else
                 torch.as_tensor(np.asarray(importance_weights), dtype=torch.float32).to(self.device).contiguous())
        K = policy.num_q
        td_stats = torch.empty(K * 2 * B, **self.tpdv) if self.use_per else None
        _lib.check(_lib.lib.ope_rddpg_critic_loss_and_grad(C.byref(cfg), C.byref(f), _lib.ptr(policy.target_actor._flat),
                                                           _lib.ptr(policy.critic._flat), _lib.ptr(policy.target_critic._flat),
                                                           _lib.ptr(u_t), _lib.ptr(w), _lib.ptr(ws), ws.numel(), _lib.ptr(gc),
                                                           _lib.ptr(td_stats), st), "ope_rddpg_critic_loss_and_grad")
        new_priorities = None
        self.gathered_priorities = None      # (what the all-reduce gathered: dist.allgather_cat(new_priorities, have=trainer.gathered_priorities))
        n_head = policy.critic.padded_numel + 4
        if self.use_per and dev_prio:
            s = td_stats.view(K, B, 2)
            nu = self.args.per_nu
            new_priorities = (((1 - nu) * s[:, :, 0] + nu * s[:, :, 1]) + self.per_eps).mean(dim=0) + self.per_eps
            if world_size > 1:      # the ranks' priorities ride on the gradient all-reduce: every rank ends up with all world x B of them, in HBM
                # (uniform contract, dist.allgather_cat: the LOCAL share's priorities are what this call returns)
                new_priorities = new_priorities.clone()
                self.gathered_priorities = opdist.priority_slots(gc, n_head, new_priorities)
        opdist.allreduce_flat_(gc if (self.use_per and dev_prio and world_size > 1) else gc[:n_head])
        cs = self._adam(policy.critic_optimizer, policy.critic.padded_numel, policy.critic._flat, policy.target_critic._flat, gc, scratch,
                        T * B * world_size, policy.critic.unused_range)
        train_info["critic_loss"], train_info["critic_grad_norm"] = cs[0], cs[1]
        if self.use_per and not dev_prio:        # r_maddpg.py:216-218: eps is added per head AND after the mean over heads
            s = td_stats.view(K, B, 2).cpu().numpy().astype(np.float32)
            nu = self.args.per_nu
            per_head = [((1 - nu) * s[k, :, 0] + nu * s[k, :, 1]).flatten() + self.per_eps for k in range(K)]
            new_priorities = np.stack(per_head).mean(axis=0) + self.per_eps
        # ---- actor (through the freshly updated critic) ----
        u_a = None
        if update_actor:
            if policy.discrete:
                u_a = override[1].to(self.device, dtype=torch.float32).contiguous() if override is not None else noise_for(policy, T, N * B, False)
                assert u_a.shape == (T, N * B, A)
            _lib.check(_lib.lib.ope_rddpg_actor_loss_and_grad(C.byref(cfg), C.byref(f), _lib.ptr(policy.actor._flat),
                                                              _lib.ptr(policy.critic._flat), _lib.ptr(u_a), _lib.ptr(ws), ws.numel(),
                                                              _lib.ptr(ga), st), "ope_rddpg_actor_loss_and_grad")
            opdist.allreduce_flat_(ga)
            as_ = self._adam(policy.actor_optimizer, policy.actor.padded_numel, policy.actor._flat, policy.target_actor._flat, ga, scratch,
                             1.0, policy.actor.unused_range)
            train_info["actor_grad_norm"], train_info["actor_loss"] = as_[1], as_[0]
        train_info["update_actor"] = update_actor
        self.num_updates[pid] += 1
        self._last = (obs, share, acts, rew, dones, dones_env, avail, u_t, u_a, w, td_stats, joint_next, keep)   # keep alive past the async launches
        return train_info, new_priorities, idxes

    def prep_training(self):
        pass

    def prep_rollout(self):
        pass
