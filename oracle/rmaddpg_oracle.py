"""CPU oracle for the recurrent MADDPG / MATD3 update path. TEST INFRASTRUCTURE ONLY (same rules as qmix_oracle.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; never the product path).

Restates in functional torch-CPU fp32, with the reference's own per-timestep loops:
    R_MADDPG.get_update_info / shared_train_policy_on_batch   offpolicy/algorithms/r_maddpg/r_maddpg.py:44-105, 114-331
    R_MADDPG_Actor / R_MADDPG_Critic                          offpolicy/algorithms/r_maddpg/algorithm/r_actor_critic.py:36-67, 93-129
    RNNBase / RNNLayer                                        offpolicy/algorithms/utils/rnn.py:19-47
    R_MADDPGPolicy.get_actions (use_target / use_gumbel)      offpolicy/algorithms/r_maddpg/algorithm/rMADDPGPolicy.py:61-131
    onehot_from_logits / gumbel_softmax                       offpolicy/utils/util.py:156-214
Uniform noise tensors are inputs. Pinned by tests/golden/rmaddpg_*.npz / rmatd3_*.npz (outputs of the real reference).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .maddpg_oracle import gumbel_hard, onehot_argmax
from .qmix_oracle import HP, gru_sequence, huber, layer_norm, mlp_trunk


def rnn_body(P, x_seq, h0):
    """RNNBase.forward: trunk -> GRU -> LayerNorm. x_seq [L, R, Din], h0 [R, 64] -> (y [L, R, 64], h_final)."""
    y, h = gru_sequence(P, mlp_trunk(P, x_seq), h0)
    return layer_norm(y, P["rnn.rnn.norm.weight"], P["rnn.rnn.norm.bias"]), h


def actor_logits(P, obs_seq, h0):
    y, h = rnn_body(P, obs_seq, h0)
    if "act.action_out.weight" in P:
        return F.linear(y, P["act.action_out.weight"], P["act.action_out.bias"]), h
    outs, i = [], 0
    while "act.action_outs.%d.weight" % i in P:      # multi-discrete (act.py:14-17): the heads' outputs side by side
        outs.append(F.linear(y, P["act.action_outs.%d.weight" % i], P["act.action_outs.%d.bias" % i]))
        i += 1
    return torch.cat(outs, dim=-1), h


def critic_q(P, K, cent, act, h0):
    """[L, R, K] Q values (head k = q_outs.k) and the final state."""
    y, h = rnn_body(P, torch.cat([cent, act], dim=-1), h0)
    qs = [F.linear(y, P["q_outs.%d.weight" % k], P["q_outs.%d.bias" % k]) for k in range(K)]
    return torch.cat(qs, dim=-1), h


class RMaddpgOracle(object):
    def __init__(self, actor, critic, actor_tgt, critic_tgt, n_agents, hp=None, td3=False, actor_update_interval=None, continuous=False,
                 head_dims=None):
        """`continuous`: Box action space (rMADDPGPolicy.py:121-129): the action is the actor's output, R_MATD3's target action adds the
        gaussian noise passed as `u_target`; no gumbel, no availability masks."""
        self.hp = hp or HP()
        self.N, self.td3, self.continuous = n_agents, td3, bool(continuous)
        # multi-discrete action space (rMADDPGPolicy.py:81-102): sizes of the sub-actions; argmax / gumbel-softmax per block, no masks
        self.head_dims = [int(x) for x in head_dims] if head_dims is not None else None
        self.K = 2 if td3 else 1
        f = lambda d: OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=torch.float32).clone()) for k, v in d.items())
        self.actor, self.critic, self.actor_tgt, self.critic_tgt = f(actor), f(critic), f(actor_tgt), f(critic_tgt)
        self.adam = {"actor": [OrderedDict(), OrderedDict(), 0], "critic": [OrderedDict(), OrderedDict(), 0]}
        self.actor_update_interval = actor_update_interval if actor_update_interval is not None else (2 if td3 else 1)
        self.num_updates = 0

    def _hard(self, lg, avail, u):
        if self.head_dims is None:
            return gumbel_hard(lg, avail, torch.as_tensor(u))
        u = torch.as_tensor(u)
        return torch.cat([gumbel_hard(l, None, uu) for l, uu in zip(lg.split(self.head_dims, dim=-1), u.split(self.head_dims, dim=-1))], dim=-1)

    def _argmax(self, lg, avail):
        if self.head_dims is None:
            return onehot_argmax(lg, avail)
        return torch.cat([onehot_argmax(l, None) for l in lg.split(self.head_dims, dim=-1)], dim=-1)

    def _adam_step(self, which, params, grads):
        hp = self.hp
        names = [k for k in params if ".fc_h." not in k]
        g_list = [grads[k] for k in names]
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in g_list)).float()
        coef = min(1.0, hp.max_grad_norm / (float(total) + 1e-6))
        st = self.adam[which]
        st[2] += 1
        b1, b2 = 0.9, 0.999
        bc1, bc2 = 1 - b1 ** st[2], 1 - b2 ** st[2]
        for k, g in zip(names, g_list):
            g = g * coef
            if hp.weight_decay != 0:      # torch.optim.Adam: L2 term added to the (already clipped) gradient of tensors that have one
                g = g + hp.weight_decay * params[k]
            if k not in st[0]:
                st[0][k], st[1][k] = torch.zeros_like(g), torch.zeros_like(g)
            m = st[0][k].mul_(b1).add_(g, alpha=1 - b1)
            v = st[1][k].mul_(b2).addcmul_(g, g, value=1 - b2)
            params[k] = params[k] - (hp.lr / bc1) * (m / (v.sqrt() / np.sqrt(bc2)).add_(hp.opti_eps))
        return float(total)

    def target_actions(self, batch, u_target=None):
        """This policy's target actions for its own agents, get_update_info (r_maddpg.py:60-97): list of n [T, B, A] tensors."""
        hp, N = self.hp, self.N
        obs, avail = batch[0], batch[6]
        obs = torch.as_tensor(np.ascontiguousarray(obs))
        avail = torch.as_tensor(np.ascontiguousarray(avail)) if avail is not None else None
        B = obs.shape[2]
        with torch.no_grad():
            s_obs = torch.cat(list(obs), dim=1)
            s_av = torch.cat(list(avail), dim=1) if avail is not None else None
            lg, _ = actor_logits(self.actor_tgt, s_obs, torch.zeros(N * B, hp.hidden_size))
            nact = ((lg + torch.as_tensor(u_target)) if u_target is not None else lg) if self.continuous else (self._hard(lg, s_av, u_target) if self.td3 else self._argmax(lg, s_av))
        return list(nact[1:].split(B, dim=1))

    def critic_loss(self, live, batch, u_target=None, weights=None, joint=None, per_agent_cent=False):
        """Returns (loss, errors list [T,B,1] per head, mask_count). batch = sample_inds 7-tuple ([N,T(+1),B,.] agent fields).
        `joint` (multi-policy updates) = (cent_act [T,B,NT*A], cent_nact [T,B,NT*A]) over ALL policies' agents.
        `per_agent_cent` = R_MADDPG.cent_train_policy_on_batch (r_maddpg.py:333-456; use_same_share_obs = False): batch[1] is
        [N, T+1, B, S]; the agents' sequences are stacked along the batch axis (lines 361-362, with the time-axis reading of the
        `[:-1]` / `[1:]` slices at 355-356 that oracle/make_golden_cent.py documents) and the joint actions, rewards and dones
        repeated per agent (376-379): the same loss over N*B episodes."""
        hp, N, K = self.hp, self.N, self.K
        obs, cent, acts, rew, dones, dones_env, avail = [torch.as_tensor(np.ascontiguousarray(x)) if x is not None else None for x in batch]
        T, B = acts.shape[1], acts.shape[2]
        H = hp.hidden_size
        if joint is not None:
            cent_act, cent_nact = joint
        else:
            with torch.no_grad():
                s_obs = torch.cat(list(obs), dim=1)                          # [T+1, N*B, D]
                s_av = torch.cat(list(avail), dim=1) if avail is not None else None
                lg, _ = actor_logits(self.actor_tgt, s_obs, torch.zeros(N * B, H))
                nact = ((lg + torch.as_tensor(u_target)) if u_target is not None else lg) if self.continuous else (self._hard(lg, s_av, u_target) if self.td3 else self._argmax(lg, s_av))
                cent_nact = torch.cat(nact[1:].split(B, dim=1), dim=-1)      # [T, B, N*A]
            cent_act = torch.cat(list(acts), dim=-1)                         # [T, B, N*A]
        if per_agent_cent:
            cent = torch.cat(list(cent), dim=1)                              # [T+1, N*B, S], column = agent * B + b
            cent_act, cent_nact = cent_act.repeat(1, N, 1), cent_nact.repeat(1, N, 1)
            rew, dones_env = rew.repeat(1, 1, N, 1), dones_env.repeat(1, N, 1)
            B = N * B
        cent_obs, cent_nobs = cent[:-1], cent[1:]
        q, _ = critic_q(live, K, cent_obs, cent_act, torch.zeros(B, H))
        with torch.no_grad():
            h = torch.zeros(B, H)
            nq = []
            for t in range(T):
                _, h = critic_q(self.critic_tgt, K, cent_obs[t:t + 1], cent_act[t:t + 1], h)
                q_side, _ = critic_q(self.critic_tgt, K, cent_nobs[t:t + 1], cent_nact[t:t + 1], h)
                nq.append(q_side[0].min(dim=-1, keepdim=True)[0])
            nq = torch.stack(nq)
            curr = torch.cat([torch.zeros(1, B, 1), dones_env[:T - 1]], dim=0)
            target = (rew[0] + hp.gamma * ((1 - dones_env) * nq)) * (1 - curr)
        errs = [q[..., k:k + 1] * (1 - curr) - target for k in range(K)]
        f = (lambda e: huber(e, hp.huber_delta)) if hp.use_huber_loss else (lambda e: e ** 2)
        cnt = (1 - curr).sum()
        if hp.use_per:
            w = torch.as_tensor(np.asarray(weights), dtype=torch.float32)
            loss = sum((f(e).sum(dim=0).flatten() * w).sum() / cnt for e in errs)
        else:
            loss = sum(f(e).sum() / cnt for e in errs)
        return loss, errs, cnt

    def actor_loss(self, live, batch, u_actor, all_acts=None, offset=0, per_agent_cent=False):
        """`all_acts` (multi-policy updates): list of every agent's buffer actions [T,B,A] in joint order; this policy's agents are
        entries offset .. offset + N - 1 (act_sequence_replace_ind_start, r_maddpg.py:66-67, 291-301).
        `per_agent_cent` (r_maddpg.py:458-556): copy i of the batch carries agent i's own centralized observation (line 544 reads
        `all_agent_cent_obs`) instead of a repeat of the shared one."""
        hp, N, K = self.hp, self.N, self.K
        obs, cent, acts, rew, dones, dones_env, avail = [torch.as_tensor(np.ascontiguousarray(x)) if x is not None else None for x in batch]
        T, B = acts.shape[1], acts.shape[2]
        H = hp.hidden_size
        s_obs = torch.cat(list(obs), dim=1)[:-1]
        s_av = torch.cat(list(avail), dim=1)[:-1] if avail is not None else None
        lg, _ = actor_logits(live, s_obs, torch.zeros(N * B, H))
        pol = lg if self.continuous else self._hard(lg, s_av, u_actor)            # [T, N*B, A]
        agent_seqs = pol.split(B, dim=1)
        stacked_obs = torch.cat(list(cent), dim=1)[:-1] if per_agent_cent else cent[:-1].repeat(1, N, 1)
        every = list(acts) if all_acts is None else list(all_acts)
        buf_joint = torch.cat(every, dim=-1).repeat(1, N, 1)             # [T, N*B, NT*A]
        rows = []
        for i in range(N):
            rows.append(torch.cat([agent_seqs[i] if a == offset + i else every[a] for a in range(len(every))], dim=-1))
        repl_joint = torch.cat(rows, dim=1)                              # copy i carries the actor's action for agent i
        h = torch.zeros(N * B, H)
        qs = []
        for t in range(T):
            q_t, _ = critic_q(self.critic, K, stacked_obs[t:t + 1], repl_joint[t:t + 1], h)
            with torch.no_grad():
                _, h = critic_q(self.critic, K, stacked_obs[t:t + 1], buf_joint[t:t + 1], h)
            qs.append(q_t[0][:, 0:1])
        qs = torch.stack(qs)                                              # [T, N*B, 1]
        dm = torch.cat([torch.cat([torch.zeros(1, B, 1), dones[i][:T - 1]], dim=0) for i in range(N)], dim=1)
        return (-(qs * (1 - dm))).sum() / (1 - dm).sum()

    def train_step(self, batch, u_target=None, u_actor=None, weights=None, soft_update=True, joint=None, all_acts=None, offset=0,
                   per_agent_cent=False):
        hp = self.hp
        update_actor = self.num_updates % self.actor_update_interval == 0
        live = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.critic.items())
        closs, errs, _ = self.critic_loss(live, batch, u_target, weights, joint=joint, per_agent_cent=per_agent_cent)
        names = [k for k in live if ".fc_h." not in k]
        cg = dict(zip(names, torch.autograd.grad(closs, [live[k] for k in names])))
        cnorm = self._adam_step("critic", self.critic, cg)
        prio = None
        if hp.use_per:
            tds = [e.abs().detach().numpy() for e in errs]
            per_head = [((1 - hp.per_nu) * td.mean(axis=0) + hp.per_nu * td.max(axis=0)).flatten() + hp.per_eps for td in tds]
            prio = np.stack(per_head).mean(axis=0) + hp.per_eps
        out = dict(critic_loss=float(closs.detach()), critic_grad_norm=cnorm, priorities=prio, update_actor=update_actor,
                   critic_grads={k: v.numpy() for k, v in cg.items()})
        if update_actor:
            la = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.actor.items())
            aloss = self.actor_loss(la, batch, u_actor, all_acts=all_acts, offset=offset, per_agent_cent=per_agent_cent)
            anames = [k for k in la if ".fc_h." not in k]
            ag = dict(zip(anames, torch.autograd.grad(aloss, [la[k] for k in anames])))
            out.update(actor_loss=float(aloss.detach()), actor_grad_norm=self._adam_step("actor", self.actor, ag),
                       actor_grads={k: v.numpy() for k, v in ag.items()})
        self.num_updates += 1
        if soft_update:
            tau = hp.tau
            for src, dst in ((self.critic, self.critic_tgt), (self.actor, self.actor_tgt)):
                for k in src:
                    dst[k] = dst[k] * (1 - tau) + src[k] * tau
        return out


class RMaddpgMultiOracle(object):
    """share_policy = False: one RMaddpgOracle per policy (own actor, critic, targets, Adam state), agents concatenated in policy
    order. `train_step(p, batches, u_targets, u_actor)` = R_MADDPG.shared_train_policy_on_batch(policy p, batch)
    (r_maddpg.py:114-331 with get_update_info 44-105 looping over every policy's target actor)."""

    def __init__(self, oracles):
        self.pol = list(oracles)
        self.offsets = np.cumsum([0] + [o.N for o in self.pol])[:-1]

    def train_step(self, p, batches, u_targets=None, u_actor=None, weights=None, soft_update=True):
        """batches[k] = policy k's sample_inds 7-tuple; u_targets[k] = its target noise (MATD3) or None."""
        nact, every = [], []
        for k, o in enumerate(self.pol):
            nact += o.target_actions(batches[k], None if u_targets is None else u_targets[k])
            every += [torch.as_tensor(np.ascontiguousarray(a)) for a in batches[k][2]]
        joint = (torch.cat(every, dim=-1), torch.cat(nact, dim=-1))
        return self.pol[p].train_step(batches[p], None, u_actor, weights, soft_update, joint=joint, all_acts=every, offset=int(self.offsets[p]))
