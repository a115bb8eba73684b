"""MLP MADDPG / MATD3 trainer on the HIP engine.

Mirror of offpolicy/algorithms/maddpg/maddpg.py:9-249 (`MADDPG`, shared-observation path
`shared_train_policy_on_batch`): critic update, then (every `actor_update_interval` updates) the actor update through
the freshly updated critic. Four C-ABI calls per policy update:

    ope_ddpg_critic_loss_and_grad -> ope_adam_step(critic) -> ope_ddpg_actor_loss_and_grad -> ope_adam_step(actor)

Upstream defects kept by default so that results are the reference's:
  A-4  critic Q heads are unregistered -> frozen (`MADDPGPolicy(frozen_q_head=True)`);
  A-5  `num_updates` is never incremented -> the actor is updated on EVERY call, also for MATD3
       (`count_updates=False`; pass True to get the intended delayed actor update).
The gumbel noise is drawn on the CPU generator in the reference's order (target noise first, then actor noise).
"""
import ctypes as C
import os

import numpy as np
import torch

from ... import _lib
from ... import dist as opdist
from .algorithm.MADDPGPolicy import sample_gumbel_uniform, gumbel_uniform_for, target_noise_for


_OPT_TAIL_DEFAULT = os.environ.get("OPE_DDPG_OPT_TAIL", "0") == "1"

class MADDPG(object):
    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=None, actor_update_interval=1, count_updates=False):
        self.args = args
        self.use_per, self.per_eps = args.use_per, args.per_eps
        self.use_huber_loss, self.huber_delta = args.use_huber_loss, args.huber_delta
        self.device = torch.device(device if device is not None else "cuda:0")
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.num_agents, self.policies, self.policy_mapping_fn = num_agents, policies, policy_mapping_fn
        self.policy_ids = sorted(list(self.policies.keys()))
        self.policy_agents = {pid: sorted([a for a in range(num_agents) if policy_mapping_fn(a) == pid]) for pid in self.policies}
        # several policies (share_policy = False; get_update_info maddpg.py:40-80): every policy has its own actor, critic and buffer;
        # the joint action is the agents' blocks in policy order
        self.multi_policy = self.policy_ids != ["policy_0"] or len(self.policy_agents["policy_0"]) != num_agents
        self.agent_offset, off = {}, 0
        for pid in self.policy_ids:
            self.agent_offset[pid] = off
            off += len(self.policy_agents[pid])
        if self.multi_policy:
            flat_agents = [a for pid in self.policy_ids for a in self.policy_agents[pid]]
            if flat_agents != list(range(num_agents)) or any(len(self.policy_agents[pid]) == 0 for pid in self.policy_ids):
                raise NotImplementedError("several policies: agents must be numbered policy by policy, every policy with at least one agent")
        # policies of different action dimensions (MPE simple_speaker_listener): the joint action is then described in columns
        # (ope_ddpg_cfg.joint_act_dim / joint_act_col / joint_acts) instead of equal agent blocks
        self.mixed_act_dims = len({self.policies[pid].output_dim for pid in self.policy_ids}) != 1
        self.joint_act_col, col = {}, 0
        for pid in self.policy_ids:
            self.joint_act_col[pid] = col
            col += len(self.policy_agents[pid]) * self.policies[pid].output_dim
        self.joint_act_dim = col
        self.num_updates = {p_id: 0 for p_id in self.policy_ids}
        self.use_same_share_obs = args.use_same_share_obs
        self.actor_update_interval = actor_update_interval
        self.count_updates = bool(count_updates)
        self._noise_ctr = None
        self._noise_seed = None
        self.device_noise = False     # True: gumbel noise drawn on the device instead of the reference's CPU generator stream
        self.fuse_soft_update = False  # True: Polyak of both target nets inside the Adam kernels; the next
                                       # policy.soft_target_updates() call is then skipped (same values, two launches fewer)
        self._ws, self._grads, self._gsq = {}, {}, {}

    def _workspace(self, policy, cfg):
        B = (cfg.batch, cfg.dims.n_agents, cfg.n_total_agents, cfg.joint_act_dim, id(policy)) if (cfg.n_total_agents or cfg.joint_act_dim) else cfg.batch
        if B not in self._ws:
            need = _lib.lib.ope_ddpg_workspace_bytes(C.byref(cfg))
            if need < 0:
                _lib.check(int(need), "ope_ddpg_workspace_bytes")
            ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib.ope_ddpg_workspace_init(C.byref(cfg), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                       "ope_ddpg_workspace_init")
            self._ws[B] = ws
            self._grads[B] = (torch.zeros(policy.critic.padded_numel + 4, **self.tpdv), torch.zeros(policy.actor.padded_numel + 4, **self.tpdv),
                              torch.zeros(int(_lib.lib.ope_adam_scratch_floats(max(policy.critic.padded_numel, policy.actor.padded_numel))), **self.tpdv))
        return self._ws[B], self._grads[B]

    def workspace_view(self, batch, name, policy_id="policy_0"):
        cfg = self.policies[policy_id].ddpg_cfg(batch)
        n = C.c_int64(0)
        off = _lib.lib.ope_ddpg_workspace_find(C.byref(cfg), name.encode(), C.byref(n))
        if off < 0:
            raise KeyError(name)
        return self._ws[batch][off:off + 4 * n.value].view(torch.float32)

    def train_policy_on_batch(self, update_policy_id, batch):
        if self.use_same_share_obs:
            return self.shared_train_policy_on_batch(update_policy_id, batch)
        return self.cent_train_policy_on_batch(update_policy_id, batch)

    def cent_train_policy_on_batch(self, update_policy_id, batch):
        """Every agent has its OWN centralized observation (use_same_share_obs = False): maddpg.py:251-419. `batch` = the 13-tuple of
        MlpReplayBuffer.sample() with cent_obs / cent_nobs [N, B, S]. Upstream the function fails on every input (it
        calls `.reshape` on the critic's LIST of Q heads); with one head (MADDPG) the intended meaning is unambiguous and is what
        runs here, pinned on outputs of the reference run with that reading (oracle/make_golden_cent.py,
        tests/golden/maddpg_cent_*.npz). With two heads (MATD3) the code gives no rule for combining them: refused.
        The critic is trained on the N*B rows (agent i's observation, the joint action), rewards / dones / importance weights
        repeated per agent (lines 279-288, 306), priorities averaged over the agents (325-326); in the actor update copy i of
        transition b carries agent i's observation (line 399). Both are the shared-observation update on a batch laid out as N*B
        transitions with the actor's copy `a` of transition (i, b) masked out unless a == i (through valid_transition): the same
        C-ABI calls, correct rather than tuned."""
        (obs_b, cent_b, act_b, rew_b, nobs_b, cent_nobs_b, dones_b, dones_env_b, valid_b, avail_b, navail_b,
         importance_weights, idxes) = batch
        pid = update_policy_id
        policy = self.policies[pid]
        if self.multi_policy:
            raise NotImplementedError("cent_train_policy_on_batch with several policies is not on the accelerated path")
        if policy.num_q != 1:
            raise NotImplementedError("cent_train_policy_on_batch with two critic heads (MATD3): upstream defines no rule for them (maddpg.py:295)")
        if not policy.discrete or policy.multidiscrete:
            raise NotImplementedError("cent_train_policy_on_batch with continuous / multi-discrete actions is not on the accelerated path (no reference fixture pins it)")
        if getattr(self.args, "use_value_active_masks", False):
            raise NotImplementedError("cent_train_policy_on_batch with use_value_active_masks: upstream weights the critic loss by the valid-transition "
                                      "mask there (maddpg.py:314-317, 327-330); the accelerated path takes the plain mean over the N*B rows")
        dev = self.device
        t = lambda x: None if x is None else torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32).to(dev)
        obs, cent, acts, rew = t(obs_b[pid]), t(cent_b[pid]), t(act_b[pid]), t(rew_b[pid])
        nobs, ncent, dones, dones_env, valid = t(nobs_b[pid]), t(cent_nobs_b[pid]), t(dones_b[pid]), t(dones_env_b[pid]), t(valid_b[pid])
        avail = t(avail_b[pid]) if avail_b is not None else None
        navail = t(navail_b[pid]) if navail_b is not None else None
        N, B, _ = obs.shape
        A = policy.output_dim
        assert cent.shape[:2] == (N, B), "per-agent centralized observations [N, B, S] expected"
        tile = lambda x: None if x is None else x.repeat(1, N, 1).contiguous()          # [N, N*B, .]: column (i, b) <- b
        eye = torch.eye(N, device=dev).repeat_interleave(B, dim=1)[..., None]           # copy a of transition (i, b) counts iff a == i
        w = importance_weights
        dev_prio = torch.is_tensor(importance_weights)      # device trees hand tensors in and take tensors back; host trees numpy
        if self.use_per:
            w = (w.to(dev, dtype=torch.float32) if dev_prio else torch.as_tensor(np.asarray(w), dtype=torch.float32).to(dev)).repeat(N)
        # the reference draws the actor's gumbel noise for the N*B real rows: same draws, repeated per copy
        u = torch.rand((N * B, A), **self.tpdv) if self.device_noise else sample_gumbel_uniform((N * B, A)).to(dev)
        self._noise_override = (None, u.view(N, B, A).repeat(1, N, 1).view(N * N * B, A).contiguous())
        vb = ({pid: tile(obs)}, {pid: cent.reshape(N * B, -1).contiguous()}, {pid: tile(acts)}, {pid: tile(rew)}, {pid: tile(nobs)},
              {pid: ncent.reshape(N * B, -1).contiguous()}, {pid: tile(dones)}, {pid: dones_env.repeat(N, 1).contiguous()},
              {pid: (tile(valid) * eye).contiguous()}, None if avail is None else {pid: tile(avail)},
              None if navail is None else {pid: tile(navail)}, w, idxes)
        info, prio, idxes = self.shared_train_policy_on_batch(pid, vb)
        if prio is not None:       # maddpg.py:325-326: mean over the agents of |TD|, + per_eps once
            eps = self.per_eps
            prio = ((prio - eps).reshape(N, B).mean(0) + eps) if torch.is_tensor(prio) else (np.asarray(prio) - eps).reshape(N, B).mean(0) + eps
            if torch.is_tensor(prio) and not dev_prio:
                prio = prio.cpu().numpy()
        return info, prio, idxes

    def _gsq_region(self, cfg, ws, name, n_opt, n_all):
        """(pointer, count) of the sum-of-squares partials the fused path left for the gradient just computed, or None.
        Only usable on the un-reduced gradient (one process). `n_opt < n_all`: the head block is frozen, take the trunk half."""
        if opdist.is_distributed():
            return None
        key = (cfg.batch, name)
        if key not in self._gsq:
            n = C.c_int64(0)
            off = _lib.lib.ope_ddpg_workspace_find(C.byref(cfg), name.encode(), C.byref(n))
            self._gsq[key] = (int(off), int(n.value)) if off >= 0 else None
        r = self._gsq[key]
        if r is None:
            return None
        return ws.data_ptr() + r[0], (r[1] // 2 if n_opt < n_all else r[1])

    def _update_in_launch(self, cfg):
        """Does the update launch carry its own optimiser step (ope_ddpg_critic_update / ope_ddpg_actor_update: slab reduction, clip, Adam
        and Polyak in the tile kernel's tail)? One process (a multi-GPU run all-reduces the gradient between the two), one shared
        policy, small networks whose workgroups are all co-resident. OFF by default: measured on MI355X the two grid barriers of the tail
        cost more than the two launches they replace (config 3: 0.1076 vs 0.0894 ms per step; csrc/ope_ddpg_tile.hip) --
        `self.update_in_launch = True` or OPE_DDPG_OPT_TAIL=1 selects it."""
        if not getattr(self, "update_in_launch", _OPT_TAIL_DEFAULT) or self.multi_policy or opdist.is_distributed():
            return False
        key = ("uil", cfg.batch)
        if key not in self._gsq:
            self._gsq[key] = bool(_lib.lib.ope_ddpg_update_ok(C.byref(cfg)))
        return self._gsq[key]

    def _opt_block(self, opt, n, flat_tgt, skip):
        opt.step_count += 1
        o = _lib.DdpgOpt()
        ac = o.adam
        ac.lr, ac.beta1, ac.beta2, ac.eps = opt.lr, opt.betas[0], opt.betas[1], opt.eps
        ac.max_grad_norm, ac.weight_decay = float(self.args.max_grad_norm), float(getattr(self.args, "weight_decay", 0.0))
        ac.skip_begin, ac.skip_end = skip
        ac.tau, ac.do_polyak = (float(self.args.tau), 1) if self.fuse_soft_update else (0.0, 0)
        ac.step, ac.qtot_denominator = opt.step_count, 1.0
        if opt.step_dev is not None:
            ac.step_counter = _lib.ptr(opt.step_dev).value
        stats = torch.empty(4, **self.tpdv)
        o.n, o.theta_tgt = int(n), _lib.ptr(flat_tgt).value
        o.adam_m, o.adam_v, o.stats_out = _lib.ptr(opt.exp_avg).value, _lib.ptr(opt.exp_avg_sq).value, _lib.ptr(stats).value
        return o, stats

    def _adam(self, opt, n, flat, flat_tgt, grad, scratch, tail, skip=(0, 0), gsq=None):
        opt.step_count += 1
        ac = _lib.AdamCfg()
        ac.lr, ac.beta1, ac.beta2, ac.eps = opt.lr, opt.betas[0], opt.betas[1], opt.eps
        ac.max_grad_norm, ac.weight_decay = float(self.args.max_grad_norm), float(getattr(self.args, "weight_decay", 0.0))
        ac.skip_begin, ac.skip_end = skip          # the unused fc_h block: torch's Adam never touches grad-less tensors
        ac.tau, ac.do_polyak = (float(self.args.tau), 1) if self.fuse_soft_update else (0.0, 0)
        ac.step, ac.qtot_denominator, ac.tail_offset = opt.step_count, 1.0, int(tail)
        if opt.step_dev is not None:
            ac.step_counter = _lib.ptr(opt.step_dev).value
        if gsq is not None:
            ac.sumsq_partials, ac.n_sumsq_partials = gsq
        stats = torch.empty(4, **self.tpdv)
        _lib.check(_lib.lib.ope_adam_step(C.byref(ac), int(n), _lib.ptr(flat), _lib.ptr(flat_tgt), _lib.ptr(opt.exp_avg),
                                          _lib.ptr(opt.exp_avg_sq), _lib.ptr(grad), _lib.ptr(scratch), _lib.ptr(stats),
                                          _lib.current_stream()), "ope_adam_step")
        return stats

    def shared_train_policy_on_batch(self, update_policy_id, batch):
        """See maddpg.py:90-249. `batch` = 13-tuple of MlpReplayBuffer.sample()."""
        (obs_b, cent_b, act_b, rew_b, nobs_b, cent_nobs_b, dones_b, dones_env_b, valid_b, avail_b, navail_b,
         importance_weights, idxes) = batch
        pid = update_policy_id
        policy = self.policies[pid]
        dev = self.device

        def f(x):       # tensors the engine's own buffer hands over pass through untouched (this runs ten times per update)
            if x is None:
                return None
            if torch.is_tensor(x) and x.dtype is torch.float32 and x.device == dev and x.is_contiguous():
                return x
            return torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous()
        obs, cent, acts, rew = f(obs_b[pid]), f(cent_b[pid]), f(act_b[pid]), f(rew_b[pid])
        nobs, ncent, dones_env, valid = f(nobs_b[pid]), f(cent_nobs_b[pid]), f(dones_env_b[pid]), f(valid_b[pid])
        avail = f(avail_b[pid]) if avail_b is not None else None
        navail = f(navail_b[pid]) if navail_b is not None else None
        N, B, D = obs.shape
        assert N == len(self.policy_agents[pid])
        cfg = policy.ddpg_cfg(B)
        if self.multi_policy:
            assert not self.device_noise, "several policies: the gumbel noise comes from the reference's CPU generator stream"
            NT, A = self.num_agents, policy.output_dim
            # joint target action: one ope_ddpg_target_actions per policy (its target actor on its agents' next observations; noise
            # drawn per policy in policy order, as get_update_info does), scattered into [B][N_total * A]
            mixed = self.mixed_act_dims
            joint_next = torch.empty(B, self.joint_act_dim if mixed else NT * A, **self.tpdv)
            keep = []
            for q in self.policy_ids:
                pol_q = self.policies[q]
                no_q = nobs if q == pid else f(nobs_b[q])
                nv_q = navail if q == pid else (f(navail_b[q]) if navail_b is not None else None)
                cq = pol_q.ddpg_cfg(B)
                cq.dims.n_agents, cq.n_total_agents, cq.agent_offset = int(no_q.shape[0]), NT, self.agent_offset[q]
                if mixed:
                    cq.n_total_agents, cq.agent_offset, cq.joint_act_dim, cq.joint_act_col = 0, 0, self.joint_act_dim, self.joint_act_col[q]
                ws_q, _ = self._workspace(pol_q, cq)
                mq = _lib.MlpBatch()
                mq.next_obs, mq.next_avail_acts = _lib.ptr(no_q).value, _lib.ptr(nv_q).value
                u_q = target_noise_for(pol_q, int(no_q.shape[0]) * B)      # per policy, in policy order: get_update_info's draws
                u_q = None if u_q is None else u_q.to(self.device)
                _lib.check(_lib.lib.ope_ddpg_target_actions(C.byref(cq), C.byref(mq), _lib.ptr(pol_q.target_actor._flat), _lib.ptr(u_q),
                                                            _lib.ptr(ws_q), ws_q.numel(), _lib.ptr(joint_next), _lib.current_stream()),
                           "ope_ddpg_target_actions")
                keep.append((no_q, nv_q, u_q))
            if mixed:      # the buffer's joint action [B][sum of widths]: every agent's block side by side, policy order
                joint_acts = torch.cat([(acts if q == pid else f(act_b[q])).permute(1, 0, 2).reshape(B, -1) for q in self.policy_ids], dim=-1).contiguous()
                assert joint_acts.shape[-1] == self.joint_act_dim
                keep.append(joint_acts)
                cfg.dims.n_agents, cfg.joint_act_dim, cfg.joint_act_col = N, self.joint_act_dim, self.joint_act_col[pid]
                cfg.joint_acts = _lib.ptr(joint_acts).value
            else:
                acts = torch.cat([acts if q == pid else f(act_b[q]) for q in self.policy_ids], dim=0).contiguous()      # [N_total][B][A]
                cfg.dims.n_agents, cfg.n_total_agents, cfg.agent_offset = N, NT, self.agent_offset[pid]
            cfg.joint_next_acts = _lib.ptr(joint_next).value
            self._keep_multi = (joint_next, keep)
        ws, (gc, ga, scratch) = self._workspace(policy, cfg)
        mb = _lib.MlpBatch()
        for k, v in dict(obs=obs, share_obs=cent, acts=acts, rewards=rew, next_obs=nobs, next_share_obs=ncent, dones_env=dones_env,
                         valid_transition=valid, avail_acts=avail, next_avail_acts=navail).items():
            setattr(mb, k, _lib.ptr(v).value)
        st = _lib.current_stream()
        train_info = {}
        update_actor = self.num_updates[pid] % self.actor_update_interval == 0
        # ---- critic ----
        if self.device_noise:      # the kernels draw the gumbel noise themselves (Philox keyed by seed, step count, row)
            draw = lambda shape: None
            opt = policy.critic_optimizer
            if opt.step_dev is not None:
                ctr = opt.step_dev
            else:
                if self._noise_ctr is None:
                    self._noise_ctr = torch.zeros(1, dtype=torch.int32, device=self.device)
                self._noise_ctr += 1
                ctr = self._noise_ctr
            if self._noise_seed is None or self._noise_seed[0] != torch.initial_seed():
                self._noise_seed = (torch.initial_seed(), (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x1234567) % (1 << 64) | 1)
            cfg.noise_seed = self._noise_seed[1]
            cfg.noise_counter = _lib.ptr(ctr).value
        else:
            # (multi-discrete: one uniform block per sub-action head, in the reference's order -- gumbel_uniform_for)
            draw = lambda shape: gumbel_uniform_for(policy, shape[0]).to(self.device)
        override, self._noise_override = getattr(self, "_noise_override", None), None      # (target noise, actor noise) to use instead of drawing
        if not policy.discrete:
            # continuous actions (MADDPGPolicy.py:107-116): no gumbel anywhere; the target action carries additive gaussian noise when the
            # policy has a target noise (MATD3), drawn here on the CPU generator in the reference's order and shape (util.py:217-218)
            assert not self.device_noise, "continuous actions: host noise"
            from .algorithm.MADDPGPolicy import gaussian_noise
            draw = lambda shape: None
        u_t = draw((N * B, policy.output_dim)) if (policy.target_noise is not None and not self.multi_policy) else None
        if not policy.discrete and policy.target_noise is not None and not self.multi_policy:
            u_t = gaussian_noise((N * B, policy.output_dim), float(policy.target_noise)).to(self.device)
        dev_prio = torch.is_tensor(importance_weights)
        w = None
        if self.use_per:
            w = (importance_weights.to(self.device, dtype=torch.float32).contiguous() if dev_prio else
                 torch.as_tensor(np.asarray(importance_weights), dtype=torch.float32).to(self.device).contiguous())
        prio = torch.empty(B, **self.tpdv) if self.use_per else None
        in_launch = self._update_in_launch(cfg)
        if in_launch:      # gradient, clip, Adam and the target's Polyak step: one launch
            ob, cs = self._opt_block(policy.critic_optimizer, policy.critic.trainable_numel, policy.target_critic._flat, policy.critic.unused_range)
            _lib.check(_lib.lib.ope_ddpg_critic_update(C.byref(cfg), C.byref(mb), _lib.ptr(policy.target_actor._flat), _lib.ptr(policy.critic._flat),
                                                       _lib.ptr(policy.target_critic._flat), _lib.ptr(u_t), _lib.ptr(w), _lib.ptr(ws), ws.numel(),
                                                       _lib.ptr(gc), _lib.ptr(prio), C.byref(ob), st), "ope_ddpg_critic_update")
        else:
            _lib.check(_lib.lib.ope_ddpg_critic_loss_and_grad(C.byref(cfg), C.byref(mb), _lib.ptr(policy.target_actor._flat),
                                                              _lib.ptr(policy.critic._flat), _lib.ptr(policy.target_critic._flat),
                                                              _lib.ptr(u_t), _lib.ptr(w), _lib.ptr(ws), ws.numel(), _lib.ptr(gc),
                                                              _lib.ptr(prio), st), "ope_ddpg_critic_loss_and_grad")
            opdist.allreduce_flat_(gc)
            cs = self._adam(policy.critic_optimizer, policy.critic.trainable_numel, policy.critic._flat, policy.target_critic._flat, gc,
                            scratch, policy.critic.padded_numel, policy.critic.unused_range,
                            self._gsq_region(cfg, ws, "gsq_critic", policy.critic.trainable_numel, policy.critic.padded_numel))
        train_info["critic_loss"], train_info["critic_grad_norm"] = cs[0], cs[1]
        new_priorities = (prio if dev_prio else prio.cpu().numpy()) if self.use_per else None
        # ---- actor ----
        if update_actor:
            u_a = draw((N * B, policy.output_dim)) if override is None else override[1]
            assert u_a is None or tuple(u_a.shape) == (N * B, policy.output_dim)
            if in_launch:
                ob, as_ = self._opt_block(policy.actor_optimizer, policy.actor.padded_numel, policy.target_actor._flat, policy.actor.unused_range)
                _lib.check(_lib.lib.ope_ddpg_actor_update(C.byref(cfg), C.byref(mb), _lib.ptr(policy.actor._flat), _lib.ptr(policy.critic._flat),
                                                          _lib.ptr(u_a), _lib.ptr(ws), ws.numel(), _lib.ptr(ga), C.byref(ob), st), "ope_ddpg_actor_update")
            else:
                _lib.check(_lib.lib.ope_ddpg_actor_loss_and_grad(C.byref(cfg), C.byref(mb), _lib.ptr(policy.actor._flat),
                                                                 _lib.ptr(policy.critic._flat), _lib.ptr(u_a), _lib.ptr(ws), ws.numel(),
                                                                 _lib.ptr(ga), st), "ope_ddpg_actor_loss_and_grad")
                opdist.allreduce_flat_(ga)
                as_ = self._adam(policy.actor_optimizer, policy.actor.padded_numel, policy.actor._flat, policy.target_actor._flat, ga,
                                 scratch, policy.actor.padded_numel, policy.actor.unused_range,
                                 self._gsq_region(cfg, ws, "gsq_actor", policy.actor.padded_numel, policy.actor.padded_numel))
            train_info["actor_loss"], train_info["actor_grad_norm"] = as_[0], as_[1]
            train_info["update_actor"] = update_actor
        elif self.fuse_soft_update:      # no actor step this time: its target still takes its Polyak step
            _lib.check(_lib.lib.ope_polyak(policy.actor.padded_numel, _lib.ptr(policy.actor._flat), _lib.ptr(policy.target_actor._flat),
                                           float(self.args.tau), st), "ope_polyak")
        if self.fuse_soft_update:
            policy._polyak_done = True
        if self.count_updates:
            self.num_updates[pid] += 1
        self._last = (obs, cent, acts, rew, nobs, ncent, dones_env, valid, avail, navail, u_t, w)
        return train_info, new_priorities, idxes

    def make_graphed_step(self, buffer, batch_size, policy_id="policy_0", device_sampling=False, steps_per_replay=1):
        if self.multi_policy:
            raise NotImplementedError("graphed step: one shared policy only")
        if not self.policies[policy_id].discrete:
            raise NotImplementedError("graphed step: discrete action spaces (the continuous target noise is drawn on the host generator)")
        return self._make_graphed_step(buffer, batch_size, policy_id, device_sampling, steps_per_replay)

    def _make_graphed_step(self, buffer, batch_size, policy_id="policy_0", device_sampling=False, steps_per_replay=1):
        """One whole update -- gather of `batch_size` transitions, critic update, actor update, soft target updates -- captured
        once as a HIP graph and replayed with one launch per step. This path is ~45 kernels of a few microseconds each:
        eagerly it is bound by launch latency and host work, not by the GPU (csrc/ope_ddpg.hip). Returns
        `step(inds) -> train_info` where `inds` are the transition indices to train on (numpy int64 [batch_size], e.g.
        np.random.choice(len(buffer), batch_size)) and train_info holds device tensors overwritten by every replay.
        Restrictions: gumbel noise drawn on the device (`device_noise`), the Adam step counters live on the device.
        Prioritized replay (`use_per`) needs the buffer's DEVICE trees (PrioritizedMlpReplayBuffer(device_tree=True)): sample
        (masses from torch.rand, tree walk, importance weights), gather, update and update_priorities are all captured;
        `step(beta)` takes the annealed beta, written to a device scalar before the replay; one process only (the ranks of a
        multi-process prioritized run exchange their priorities through the host: that step stays eager).
        In a multi-process run (one rank per GPU) the two gradient all-reduces are
        captured with the rest: that needs the one-shot xGMI exchange (dist.setup_fast_allreduce verified it; its call counter
        lives on the device, ope_allreduce_flat_dev) -- with the RCCL fallback the step stays eager. Every rank replays its own
        graph on its share of the batch (`batch_size` = the local share); device-drawn indices use a per-rank stream.
        `device_sampling=True`: the batch indices are drawn on the device too (buffer.sample(batch_size) with the uniform draw
        inside the gather kernel, MlpPolicyBuffer.sample_device): `step()` takes no argument, a replay involves no host data at
        all, and train_info["indices"] holds the drawn indices. With device sampling a replay may also hold several CONSECUTIVE
        training steps (`steps_per_replay`; the counters that key indices, noise and Adam's bias correction advance on the device
        between them), which amortises the graph-launch latency; train_info is then that of the last step of the replay."""
        steps_per_replay = int(steps_per_replay)
        assert steps_per_replay == 1 or device_sampling or self.use_per, "several steps per replay need the indices drawn on the device"
        if self.use_per and (opdist.is_distributed() or not getattr(buffer, "device_tree", False)):
            raise NotImplementedError("graphed prioritized step: one process, device trees (PrioritizedMlpReplayBuffer(device_tree=True))")
        pid = policy_id
        policy, pbuf = self.policies[pid], buffer.policy_buffers[pid]
        if opdist.is_distributed() and not opdist.graph_safe_allreduce(max(policy.critic.padded_numel, policy.actor.padded_numel) + 4):
            raise NotImplementedError("graphed step at world > 1 needs the one-shot all-reduce with slots that hold the gradient vectors (%s)" % opdist.allreduce_backend())
        B = int(batch_size)
        # `update_actor` is decided on the HOST (num_updates % actor_update_interval, maddpg.py:100) while the graph is being
        # captured, so the captured launch sequence fixes the actor cadence: it is the eager one only if a replay holds whole
        # periods of it and starts at the beginning of one. (count_updates=False = the reference's defect A-5: every step updates.)
        if self.count_updates and self.actor_update_interval > 1:
            assert steps_per_replay % self.actor_update_interval == 0 and self.num_updates[pid] % self.actor_update_interval == 0, \
                "graphed step with a delayed actor: steps_per_replay must be a multiple of actor_update_interval (%d) and the " \
                "update count aligned to it" % self.actor_update_interval
        self.device_noise = True
        self.fuse_soft_update = True
        for opt in (policy.critic_optimizer, policy.actor_optimizer):
            opt.step_dev = torch.tensor([opt.step_count, 0], dtype=torch.int32, device=self.device)    # [count, ticket]
        static_inds = torch.zeros(B, dtype=torch.int64, device=self.device)

        sample_seed = ((torch.initial_seed() + 0x632BE59BD9B4E019 * opdist.world()[0]) * 0xD1342543DE82EF95 + 0x9E3779B9) % (1 << 64) | 1   # per rank

        beta_dev = torch.ones(1, dtype=torch.float64, device=self.device) if self.use_per else None

        def body():
            if self.use_per:       # sample by priority, train with the importance weights, write the new priorities back: no host data
                batch = buffer.sample_device(B, beta_dev, p_id=pid)
                info, prio, drawn = self.shared_train_policy_on_batch(pid, batch)
                buffer.update_priorities(drawn, prio, p_id=pid)
                policy.soft_target_updates()
                info["indices"], info["priorities"] = drawn, prio
                return info
            if device_sampling:
                s, drawn = pbuf.sample_device(B, sample_seed, counter=policy.critic_optimizer.step_dev)
            else:
                s, drawn = pbuf.sample_inds(static_inds), static_inds
            info, _, _ = self.shared_train_policy_on_batch(pid, tuple({pid: x} for x in s) + (None, None))
            policy.soft_target_updates()
            info["indices"] = drawn
            return info
        # The warm-up below really trains (two critic + actor + Polyak updates on a throw-away batch): snapshot networks,
        # targets, Adam moments and step counters and restore them afterwards, so that building a graphed step leaves the
        # trainer where an eager run would be. (Sticky by design: device_noise and fuse_soft_update stay switched on.)
        nets = (policy.actor, policy.critic, policy.target_actor, policy.target_critic)
        opts = (policy.critic_optimizer, policy.actor_optimizer)
        snap_nets = [m._flat.clone() for m in nets]
        snap_opts = [(o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step_dev.clone(), o.step_count) for o in opts]
        snap_misc = (dict(self.num_updates), getattr(policy, "_polyak_done", False), np.random.get_state())
        snap_per = (buffer._dtrees[pid].trees.clone(), torch.cuda.get_rng_state(self.device)) if self.use_per else None
        side = torch.cuda.Stream(device=self.device)     # warm-up off the capture: workspaces, allocator pools, lazy init
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            static_inds.copy_(torch.from_numpy(np.random.choice(len(buffer), B)).to(self.device))
            for _ in range(2):
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        for m, v in zip(nets, snap_nets):      # the (frozen) Q heads are views into _flat: restored with it
            m._flat.copy_(v)
        for o, (m1, m2, sd, sc) in zip(opts, snap_opts):
            o.exp_avg.copy_(m1); o.exp_avg_sq.copy_(m2); o.step_dev.copy_(sd); o.step_count = sc
        self.num_updates.update(snap_misc[0])
        policy._polyak_done = snap_misc[1]
        np.random.set_state(snap_misc[2])
        if snap_per is not None:      # the warm-up updates rewrote priorities and drew masses
            buffer._dtrees[pid].trees.copy_(snap_per[0])
            torch.cuda.set_rng_state(snap_per[1], self.device)
        torch.cuda.synchronize(self.device)
        count0 = dict(self.num_updates)
        actor_host0 = policy.actor_optimizer.step_count
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(steps_per_replay):
                info = body()
        actor_steps_per_replay = policy.actor_optimizer.step_count - actor_host0     # actor updates one replay performs
        self.num_updates.update(count0)        # capture ran the host code but no kernels
        for opt in (policy.critic_optimizer, policy.actor_optimizer):
            opt.step_count = int(opt.step_dev[0].item())
        ring = [(torch.empty(B, dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        state = {"k": 0, "used": [False] * 8}
        # at world > 1 the graph holds launches of the one-shot exchange, i.e. raw pointers into its local and peer buffers: it may only be
        # replayed while that exchange is the one in use (dist.disable_fast_allreduce retires it)
        ar_gen = opdist.note_graph_capture() if opdist.is_distributed() else None

        def check_exchange():
            if ar_gen is not None and opdist.fast_generation() != ar_gen:
                raise RuntimeError("this graphed step captured the one-shot all-reduce, which has since been disabled (%s): build a new "
                                   "graphed step or train eagerly" % opdist.allreduce_backend())

        def step_sampled(beta=None):
            check_exchange()
            if self.use_per:
                assert beta is not None and beta > 0, "prioritized graphed step: pass the current beta"
                beta_dev.fill_(float(beta))
            graph.replay()
            policy.critic_optimizer.step_count += steps_per_replay
            policy.actor_optimizer.step_count += actor_steps_per_replay
            if self.count_updates:
                self.num_updates[pid] += steps_per_replay
            return info

        def step(inds):
            check_exchange()
            k = state["k"]
            state["k"] = (k + 1) % 8
            host, ev = ring[k]
            if state["used"][k]:
                ev.synchronize()
            host.copy_(torch.from_numpy(np.asarray(inds, dtype=np.int64)))
            static_inds.copy_(host, non_blocking=True)
            ev.record()
            state["used"][k] = True
            graph.replay()
            policy.critic_optimizer.step_count += 1
            policy.actor_optimizer.step_count += actor_steps_per_replay
            if self.count_updates:
                self.num_updates[pid] += 1
            return info
        self._graph = (graph, static_inds, ring)      # keep alive
        return step_sampled if (device_sampling or self.use_per) else step

    def prep_training(self):
        pass

    def prep_rollout(self):
        pass
