"""CPU oracle for the MLP MADDPG / MATD3 update path. TEST INFRASTRUCTURE ONLY (same rules as qmix_oracle.py).

Restates in functional torch-CPU fp32:
    MADDPG.get_update_info / shared_train_policy_on_batch   offpolicy/algorithms/maddpg/maddpg.py:38-81, 90-249
    MADDPG_Actor / MADDPG_Critic                            offpolicy/algorithms/maddpg/algorithm/actor_critic.py:28-41, 69-87
    MADDPGPolicy.get_actions (use_target / use_gumbel)      offpolicy/algorithms/maddpg/algorithm/MADDPGPolicy.py:63-105
    onehot_from_logits / gumbel_softmax                     offpolicy/utils/util.py:156-214
    MADDPGPolicy.soft_target_updates                        MADDPGPolicy.py:141-145
including the upstream behaviours a drop-in must keep: the critic's Q heads are frozen and the
target critic has its own heads; the actor is updated on every call. Uniform noise tensors are inputs.
Pinned by tests/golden/maddpg_*.npz / matd3_*.npz (outputs of the real reference).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .mqmix_oracle import mlp_base
from .qmix_oracle import huber, HP


def gumbel_hard(logits, avail, u):
    y = logits + (-torch.log(-torch.log(u + 1e-20) + 1e-20))
    if avail is not None:
        y = y.masked_fill(avail == 0, -1e10)
    y = F.softmax(y, dim=-1)
    y_hard = (y == y.max(-1, keepdim=True)[0]).float()
    return (y_hard - y).detach() + y


def onehot_argmax(logits, avail):
    if avail is not None:
        logits = logits.masked_fill(avail == 0, -1e10)
    return (logits == logits.max(-1, keepdim=True)[0]).float()


class MaddpgOracle(object):
    def __init__(self, actor, critic, critic_heads, actor_tgt, critic_tgt, critic_heads_tgt, n_agents, hp=None, td3=False, continuous=False,
                 head_dims=None):
        """actor/critic: {name: array} trunks (+ 'act.action_out.*' for the actor); critic_heads: (W [K,64], b [K]).
        `continuous`: Box action space (MADDPGPolicy.py:107-116): the action is the actor's output; the target action of MATD3 adds
        the gaussian noise passed as `u_target` (gaussian_noise(shape, target_noise), util.py:217-218); no gumbel, no masks."""
        self.hp = hp or HP()
        self.N, self.td3, self.continuous = n_agents, td3, bool(continuous)
        # multi-discrete action space (MADDPGPolicy.py:73-92): `head_dims` = sizes of the sub-actions; the actor has one Linear head per
        # sub-action (act.action_outs.{i}.*, act.py:14-17), the action is the concatenation of their one-hot / gumbel-softmax blocks
        self.head_dims = [int(x) for x in head_dims] if head_dims is not None else None
        f = lambda d: OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=torch.float32).clone()) for k, v in d.items())
        h = lambda wb: (torch.as_tensor(np.asarray(wb[0]), dtype=torch.float32).clone(), torch.as_tensor(np.asarray(wb[1]), dtype=torch.float32).clone())
        self.actor, self.critic, self.actor_tgt, self.critic_tgt = f(actor), f(critic), f(actor_tgt), f(critic_tgt)
        self.heads, self.heads_tgt = h(critic_heads), h(critic_heads_tgt)
        self.adam = {"actor": [OrderedDict(), OrderedDict(), 0], "critic": [OrderedDict(), OrderedDict(), 0]}

    @staticmethod
    def actor_logits(P, x):
        a2 = mlp_base(P, x)
        if "act.action_out.weight" in P:
            return F.linear(a2, P["act.action_out.weight"], P["act.action_out.bias"])
        outs, i = [], 0
        while "act.action_outs.%d.weight" % i in P:      # multi-discrete: the heads' outputs side by side
            outs.append(F.linear(a2, P["act.action_outs.%d.weight" % i], P["act.action_outs.%d.bias" % i]))
            i += 1
        return torch.cat(outs, dim=-1)

    def _hard(self, lg, avail, u):
        """hard gumbel-softmax of the action vector: per sub-action block for a multi-discrete space (no availability masks there)."""
        if self.head_dims is None:
            return gumbel_hard(lg, avail, torch.as_tensor(u))
        u = torch.as_tensor(u)
        return torch.cat([gumbel_hard(l, None, uu) for l, uu in zip(lg.split(self.head_dims, dim=-1), u.split(self.head_dims, dim=-1))], dim=-1)

    def _argmax(self, lg, avail):
        if self.head_dims is None:
            return onehot_argmax(lg, avail)
        return torch.cat([onehot_argmax(l, None) for l in lg.split(self.head_dims, dim=-1)], dim=-1)

    @staticmethod
    def critic_q(P, heads, cent, joint):
        a2 = mlp_base(P, torch.cat([cent, joint], dim=1))
        return F.linear(a2, heads[0], heads[1])                       # [rows, K]

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
        """This policy's part of get_update_info (maddpg.py:56-74): its target actor on its agents' next observations ->
        list of per-agent next actions [B, A] (one-hot argmax, or hard gumbel-softmax for MATD3)."""
        nobs, navail = batch[4], batch[10]
        nobs = torch.as_tensor(np.ascontiguousarray(nobs))
        B = nobs.shape[1]
        s_nav = torch.cat(list(torch.as_tensor(np.ascontiguousarray(navail))), 0) if navail is not None else None
        with torch.no_grad():
            lg = self.actor_logits(self.actor_tgt, torch.cat(list(nobs), 0))
            if self.continuous:
                nact = lg + torch.as_tensor(u_target) if u_target is not None else lg
            else:
                nact = self._hard(lg, s_nav, u_target) if self.td3 else self._argmax(lg, s_nav)
        return list(nact.split(B, dim=0))

    def train_step(self, batch, u_target=None, u_actor=None, weights=None, soft_update=True, joint=None, all_acts=None, offset=0,
                   per_agent_cent=False):
        """`joint` / `all_acts` / `offset` (several policies, maddpg.py:40-80): joint = (buffer actions of ALL agents [B, NT*A], next
        actions of ALL agents' target actors [B, NT*A]); all_acts = the per-agent buffer actions; offset = this policy's first agent.
        `per_agent_cent` = MADDPG.cent_train_policy_on_batch (maddpg.py:251-419; use_same_share_obs = False): batch[1] / batch[5] are
        [N, B, S] -- every agent's own centralized observation -- and the critic is trained on the N*B rows (agent i's observation,
        the joint action; lines 279-286), rewards / dones / importance weights repeated per agent (287-288, 306), priorities
        averaged over the agents (325-326); the actor's copy i is evaluated on agent i's observation (line 399). Upstream the
        function crashes on the critic's list of heads; the pinned semantics are those of oracle/make_golden_cent.py's patch: head 0
        (one head: MADDPG)."""
        hp, N = self.hp, self.N
        obs, cent, acts, rew, nobs, ncent, dones, dones_env, valid, avail, navail = [
            torch.as_tensor(np.ascontiguousarray(x)) if x is not None else None for x in batch]
        B = obs.shape[1]
        s_obs, s_nobs = torch.cat(list(obs), 0), torch.cat(list(nobs), 0)
        s_av = torch.cat(list(avail), 0) if avail is not None else None
        s_nav = torch.cat(list(navail), 0) if navail is not None else None
        with torch.no_grad():
            if joint is None:
                lg = self.actor_logits(self.actor_tgt, s_nobs)
                if self.continuous:
                    nact = lg + torch.as_tensor(u_target) if u_target is not None else lg
                else:
                    nact = self._hard(lg, s_nav, u_target) if self.td3 else self._argmax(lg, s_nav)
                cent_nact = torch.cat(nact.split(B, dim=0), dim=-1)
            else:
                cent_nact = joint[1]
            rep = N if per_agent_cent else 1
            if per_agent_cent:
                assert not self.td3 and joint is None, "cent variant: one critic head, one policy (see make_golden_cent.py)"
                cent, ncent = torch.cat(list(cent), 0), torch.cat(list(ncent), 0)            # [N*B, S], row = agent * B + b
            nq = self.critic_q(self.critic_tgt, self.heads_tgt, ncent, cent_nact.repeat(rep, 1)).min(dim=-1, keepdim=True)[0]
            target = rew[0].view(-1, 1).repeat(rep, 1) + hp.gamma * (1 - dones_env.view(-1, 1).repeat(rep, 1)) * nq
        cent_act = torch.cat(list(acts), dim=-1) if joint is None else joint[0]
        live = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.critic.items())
        q = self.critic_q(live, self.heads, cent, cent_act.repeat(rep, 1))
        errs = [target - q[:, k:k + 1] for k in range(q.shape[1])]
        f = (lambda e: huber(e, hp.huber_delta)) if hp.use_huber_loss else (lambda e: e ** 2)
        if hp.use_per:
            w = torch.as_tensor(np.asarray(weights), dtype=torch.float32).repeat(rep)
            closs = sum((f(e).flatten() * w).mean() for e in errs)
            td = np.stack([e.abs().detach().numpy().flatten() for e in errs]).mean(axis=0)
            prio = (np.mean(np.split(td, rep), axis=0) if rep > 1 else td) + hp.per_eps
        else:
            closs = sum(f(e).mean() for e in errs)
            prio = None
        names = [k for k in live if ".fc_h." not in k]
        cg = dict(zip(names, torch.autograd.grad(closs, [live[k] for k in names])))
        cnorm = self._adam_step("critic", self.critic, cg)
        # ---- actor (through the UPDATED critic, whose parameters are frozen here) ----
        la = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.actor.items())
        pol = self.actor_logits(la, s_obs) if self.continuous else self._hard(self.actor_logits(la, s_obs), s_av, u_actor)
        agent_acts = pol.split(B, dim=0)
        rows = []
        every = list(acts) if all_acts is None else list(all_acts)
        for i in range(N):
            parts = [agent_acts[i] if a == offset + i else every[a] for a in range(len(every))]
            rows.append(torch.cat(parts, dim=-1))
        joint = torch.cat(rows, dim=0)
        q1 = self.critic_q(self.critic, self.heads, cent if per_agent_cent else cent.repeat(N, 1), joint)[:, 0:1]
        vmask = torch.cat(list(valid), dim=0)
        aloss = -(q1 * vmask).sum() / vmask.sum()
        anames = [k for k in la if ".fc_h." not in k]
        ag = dict(zip(anames, torch.autograd.grad(aloss, [la[k] for k in anames])))
        anorm = self._adam_step("actor", self.actor, ag)
        if soft_update:
            tau = hp.tau
            for src, dst in ((self.critic, self.critic_tgt), (self.actor, self.actor_tgt)):
                for k in src:
                    dst[k] = dst[k] * (1 - tau) + src[k] * tau
        return dict(critic_loss=float(closs.detach()), critic_grad_norm=cnorm, actor_loss=float(aloss.detach()), actor_grad_norm=anorm,
                    priorities=prio, critic_grads={k: v.numpy() for k, v in cg.items()}, actor_grads={k: v.numpy() for k, v in ag.items()})


class MaddpgMultiOracle(object):
    """share_policy = False: one MaddpgOracle per policy (own actor, critic, targets, Adam state), agents concatenated in policy order.
    `train_step(p, batches, u_targets, u_actor)` = MADDPG.shared_train_policy_on_batch(policy p, batch) (maddpg.py:90-249 with
    get_update_info 40-80 looping over every policy's target actor). Pinned by tests/golden/{maddpg_multi,matd3_multi_per}.npz."""

    def __init__(self, oracles):
        self.pol = list(oracles)
        self.offsets = np.cumsum([0] + [o.N for o in self.pol])[:-1]

    def train_step(self, p, batches, u_targets=None, u_actor=None, weights=None, soft_update=False):
        """batches[k] = policy k's sample_inds 11-tuple; u_targets[k] = its target noise (MATD3) or None."""
        nact, every = [], []
        for k, o in enumerate(self.pol):
            nact += o.target_actions(batches[k], None if u_targets is None else u_targets[k])
            every += [torch.as_tensor(np.ascontiguousarray(a)) for a in batches[k][2]]
        joint = (torch.cat(every, dim=-1), torch.cat(nact, dim=-1))
        return self.pol[p].train_step(batches[p], None, u_actor, weights, soft_update, joint=joint, all_acts=every, offset=int(self.offsets[p]))

    def soft_target_updates(self):
        for o in self.pol:
            tau = o.hp.tau
            for src, dst in ((o.critic, o.critic_tgt), (o.actor, o.actor_tgt)):
                for k in src:
                    dst[k] = dst[k] * (1 - tau) + src[k] * tau
