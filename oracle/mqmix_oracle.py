"""CPU oracle for the MLP (single-transition) QMIX / VDN update path.

TEST INFRASTRUCTURE ONLY (same rules as qmix_oracle.py). Restates, in functional torch-CPU fp32:
    MlpPolicyBuffer.sample_inds        offpolicy/utils/mlp_buffer.py:213-257
    M_QMix.train_policy_on_batch       offpolicy/algorithms/mqmix/mqmix.py:68-218
    AgentQFunction (MLP)               offpolicy/algorithms/mqmix/algorithm/agent_q_function.py:28-41
    M_QMixer / M_VDNMixer              offpolicy/algorithms/mqmix/algorithm/mq_mixer.py:80-125, mvdn/algorithm/mvdn_mixer.py:30-39
Pinned by tests/golden/mqmix_*.npz / mvdn_*.npz (outputs of the real reference; mvdn with the A-1 patch).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .qmix_oracle import layer_norm, qmixer_forward, huber, HP, q_head_dims


def mlp_agent_param_shapes(obs_dim, act_dim, hidden=64, feature_norm=True):
    H = hidden
    return OrderedDict(([("mlp.feature_norm.weight", (obs_dim,)), ("mlp.feature_norm.bias", (obs_dim,))] if feature_norm else []) + [   # mlp.py:60-62
        ("mlp.mlp.fc1.0.weight", (H, obs_dim)), ("mlp.mlp.fc1.0.bias", (H,)),
        ("mlp.mlp.fc1.2.weight", (H,)), ("mlp.mlp.fc1.2.bias", (H,)),
        ("mlp.mlp.fc_h.0.weight", (H, H)), ("mlp.mlp.fc_h.0.bias", (H,)),
        ("mlp.mlp.fc_h.2.weight", (H,)), ("mlp.mlp.fc_h.2.bias", (H,)),
        ("mlp.mlp.fc2.0.0.weight", (H, H)), ("mlp.mlp.fc2.0.0.bias", (H,)),
        ("mlp.mlp.fc2.0.2.weight", (H,)), ("mlp.mlp.fc2.0.2.bias", (H,)),
        ("q.action_out.weight", (act_dim, H)), ("q.action_out.bias", (act_dim,)),
    ])


def mlp_base(P, x, prefix="mlp.", use_relu=True):
    """MLPBase.forward, layer_N = 1 (algorithms/utils/mlp.py:76-89). The input LayerNorm only with use_feature_normalization (mlp.py:60-62,
    77-78: absent from the parameter dict otherwise); active_func = [nn.Tanh(), nn.ReLU()][use_ReLU] (mlp.py:9-12)."""
    act = F.relu if use_relu else torch.tanh
    if prefix + "feature_norm.weight" in P:
        x = layer_norm(x, P[prefix + "feature_norm.weight"], P[prefix + "feature_norm.bias"])
    x = layer_norm(act(F.linear(x, P[prefix + "mlp.fc1.0.weight"], P[prefix + "mlp.fc1.0.bias"])),
                   P[prefix + "mlp.fc1.2.weight"], P[prefix + "mlp.fc1.2.bias"])
    x = layer_norm(act(F.linear(x, P[prefix + "mlp.fc2.0.0.weight"], P[prefix + "mlp.fc2.0.0.bias"])),
                   P[prefix + "mlp.fc2.0.2.weight"], P[prefix + "mlp.fc2.0.2.bias"])
    return x


def mlp_agent_q(P, x, use_relu=True):
    y = mlp_base(P, x, use_relu=use_relu)
    if "q.action_out.weight" in P:
        return F.linear(y, P["q.action_out.weight"], P["q.action_out.bias"])
    # MultiDiscrete action space (act.py:14-17, 28-33): one Linear head per sub-action, their q blocks side by side
    return torch.cat([F.linear(y, P["q.action_outs.%d.weight" % i], P["q.action_outs.%d.bias" % i]) for i in range(len(q_head_dims(P)))], dim=-1)


def sample_inds(store, inds):
    """MlpPolicyBuffer.sample_inds (same-share, no reward normalisation): [N, B, dim] casts of store[inds]."""
    cast = lambda x: x.transpose(1, 0, 2)
    g = lambda k: store[k][inds]
    return (cast(g("obs")), g("share_obs"), cast(g("acts")), cast(g("rewards")), cast(g("next_obs")), g("next_share_obs"),
            cast(g("dones")), g("dones_env"), cast(g("valid_transition")), cast(g("avail_acts")), cast(g("next_avail_acts")))


def mlp_agent_qs(hp, agent, agent_tgt, obs, acts, nobs, navail):
    """The per-policy part of M_QMix.train_policy_on_batch (mqmix.py:95-174): q values of the actions taken [B, n] (with grad) and the
    target network's next-step q values at the greedy actions [B, n] (no grad) for ONE policy's agents (torch tensors, [n, B, .])."""
    n, B, D = obs.shape
    relu = bool(getattr(hp, "use_relu", True))
    s_obs, s_nobs, s_act = torch.cat(list(obs), 0), torch.cat(list(nobs), 0), torch.cat(list(acts), 0)
    s_nav = torch.cat(list(navail), 0) if navail is not None else None
    q_all = mlp_agent_q(agent, s_obs, relu)
    heads = q_head_dims(agent)
    if len(heads) > 1:
        # MultiDiscrete (mqmix.py:116-130, 144-155; mQMixPolicy.py:47-55): chosen / greedy / target q per sub-action head, one mixer input per
        # (agent, sub-action), agent-major; double-Q only (upstream's other branch reduces over the heads and then fails in the mixer), no masks
        assert s_nav is None and hp.use_double_q
        q_taken = torch.cat([torch.gather(qb, 1, ab.max(dim=-1)[1].unsqueeze(-1)) for qb, ab in zip(q_all.split(heads, -1), s_act.split(heads, -1))], dim=-1)
        agent_q = torch.cat(q_taken.split(B, dim=0), dim=-1)                      # [B, n * heads]
        with torch.no_grad():
            nq_blocks = mlp_agent_q(agent, s_nobs, relu).detach().split(heads, -1)
            tq_blocks = mlp_agent_q(agent_tgt, s_nobs, relu).split(heads, -1)
            tq = torch.cat([torch.gather(tb, 1, nb.max(dim=-1)[1].unsqueeze(-1)) for tb, nb in zip(tq_blocks, nq_blocks)], dim=-1)
            agent_nq = torch.cat(tq.split(B, dim=0), dim=-1)
        return agent_q, agent_nq
    q_taken = torch.gather(q_all, 1, s_act.max(dim=-1)[1].unsqueeze(-1))
    agent_q = torch.cat(q_taken.split(B, dim=0), dim=-1)                          # [B, n]
    with torch.no_grad():
        if hp.use_double_q:
            nq = mlp_agent_q(agent, s_nobs, relu).detach().clone()
            if s_nav is not None:
                nq[s_nav == 0.0] = -1e10
            nact = nq.max(dim=-1)[1]
            tq = torch.gather(mlp_agent_q(agent_tgt, s_nobs, relu), 1, nact.unsqueeze(-1))
        else:
            tqa = mlp_agent_q(agent_tgt, s_nobs, relu).clone()
            if s_nav is not None:
                tqa[s_nav == 0.0] = -1e10
            tq = tqa.max(dim=-1)[0].unsqueeze(-1)
        agent_nq = torch.cat(tq.split(B, dim=0), dim=-1)
    return agent_q, agent_nq


class MQMixOracle(object):
    def __init__(self, agent_params, mixer_params, n_agents, hp=None):
        self.hp = hp or HP()
        self.n_agents = n_agents
        f = lambda d: OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=torch.float32).clone()) for k, v in d.items())
        self.agent = f(agent_params)
        self.mixer = f(mixer_params) if mixer_params is not None else OrderedDict()
        self.agent_tgt, self.mixer_tgt = f(self.agent), f(self.mixer)
        self.adam_m, self.adam_v, self.adam_t = OrderedDict(), OrderedDict(), 0

    def _trainable(self):
        return [("agent", k) for k in self.agent if ".fc_h." not in k] + [("mixer", k) for k in self.mixer]

    def loss(self, agent, mixer, batch, weights=None):
        hp = self.hp
        obs, cent, acts, rew, nobs, ncent, dones, dones_env, valid, avail, navail = [
            torch.as_tensor(np.ascontiguousarray(x)) if x is not None else None for x in batch]
        agent_q, agent_nq = mlp_agent_qs(hp, agent, self.agent_tgt, obs, acts, nobs, navail)
        return self._mix_and_loss(mixer, agent_q, agent_nq, cent, ncent, rew, dones_env, weights)

    def _mix_and_loss(self, mixer, agent_q, agent_nq, cent, ncent, rew, dones_env, weights):
        """mqmix.py:176-205: mixer (or VDN sum) over all agents' q values, TD target, mean loss."""
        hp = self.hp
        if hp.vdn:
            q_tot = agent_q.sum(dim=-1, keepdim=True)
            nq_tot = agent_nq.sum(dim=-1, keepdim=True)
        else:
            q_tot = qmixer_forward(mixer, agent_q[None], cent[None], agent_q.shape[-1], hp.mixer_hidden_dim)[0]      # inputs: one per agent (x sub-action)
            with torch.no_grad():
                nq_tot = qmixer_forward(self.mixer_tgt, agent_nq[None], ncent[None], agent_nq.shape[-1], hp.mixer_hidden_dim)[0]
        target = rew[0] + (1 - dones_env) * hp.gamma * nq_tot
        err = q_tot - target.detach()
        el = huber(err, hp.huber_delta) if hp.use_huber_loss else err ** 2
        if hp.use_per:
            loss = (el.flatten() * torch.as_tensor(np.asarray(weights), dtype=torch.float32)).mean()
        else:
            loss = el.mean()
        return loss, err, q_tot

    def train_step(self, batch, weights=None, soft_update=True):
        hp = self.hp
        la = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.agent.items())
        lm = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.mixer.items())
        loss, err, q_tot = self.loss(la, lm, batch, weights)
        names = self._trainable()
        tensors = [la[k] if g == "agent" else lm[k] for g, k in names]
        grads = torch.autograd.grad(loss, tensors, allow_unused=True)
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None)).float()
        coef = min(1.0, hp.max_grad_norm / (float(total) + 1e-6))
        self.adam_t += 1
        b1, b2 = 0.9, 0.999
        bc1, bc2 = 1 - b1 ** self.adam_t, 1 - b2 ** self.adam_t
        for (grp, k), g in zip(names, grads):
            if g is None:
                continue
            g = g * coef
            key = grp + "/" + k
            if key not in self.adam_m:
                self.adam_m[key], self.adam_v[key] = torch.zeros_like(g), torch.zeros_like(g)
            m = self.adam_m[key].mul_(b1).add_(g, alpha=1 - b1)
            v = self.adam_v[key].mul_(b2).addcmul_(g, g, value=1 - b2)
            dst = self.agent if grp == "agent" else self.mixer
            dst[k] = dst[k] - (hp.lr / bc1) * (m / (v.sqrt() / np.sqrt(bc2)).add_(hp.opti_eps))
        out = dict(loss=float(loss.detach()), grad_norm=float(total), Q_tot=float(q_tot.detach().mean()),
                   priorities=(err.abs().detach().numpy().flatten() + hp.per_eps) if hp.use_per else None,
                   grads=OrderedDict((grp + "/" + k, g.detach().numpy() if g is not None else None) for (grp, k), g in zip(names, grads)))
        if soft_update:
            tau = hp.tau
            for src, dst in ((self.agent, self.agent_tgt), (self.mixer, self.mixer_tgt)):
                for k in src:
                    dst[k] = dst[k] * (1.0 - tau) + src[k] * tau
        return out


class MQMixMultiOracle(MQMixOracle):
    """Several policies under one mixer (share_policy = False: mqmix.py:95-178), policies of different observation width / action count /
    agent count. Policy i's parameters live under "p{i}/<name>"; one Adam, one clip norm (mqmix.py:58-63,208-210). Pinned by
    tests/golden/{mqmix,mvdn}_multi*.npz (oracle/make_golden_multi.py)."""

    def __init__(self, agent_params_list, mixer_params, n_agents_total, hp=None):
        merged = OrderedDict(("p%d/%s" % (i, k), v) for i, P in enumerate(agent_params_list) for k, v in P.items())
        super().__init__(merged, mixer_params, n_agents_total, hp)

    @staticmethod
    def _of(params, i):
        pre = "p%d/" % i
        return OrderedDict((k[len(pre):], v) for k, v in params.items() if k.startswith(pre))

    def loss(self, agent, mixer, batch, weights=None):
        """`batch`: list of the policies' 11-tuples. Centralized observations and dones_env are the FIRST policy's (mqmix.py:78-86), the
        reward the LAST policy's agent 0 (mqmix.py:100,181)."""
        tb = [[torch.as_tensor(np.ascontiguousarray(x)) if x is not None else None for x in b] for b in batch]
        qs, nqs = [], []
        for i, (obs, cent, acts, rew, nobs, ncent, dones, dones_env, valid, avail, navail) in enumerate(tb):
            q, nq = mlp_agent_qs(self.hp, self._of(agent, i), self._of(self.agent_tgt, i), obs, acts, nobs, navail)
            qs.append(q)
            nqs.append(nq)
        return self._mix_and_loss(mixer, torch.cat(qs, dim=-1), torch.cat(nqs, dim=-1), tb[0][1], tb[0][5], tb[-1][3], tb[0][7], weights)
