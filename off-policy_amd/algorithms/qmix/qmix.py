"""Recurrent QMIX / VDN trainer on the HIP engine.

Mirror of offpolicy/algorithms/qmix/qmix.py:10-232 (`QMix`): same constructor, `train_policy_on_batch`,
`soft_target_updates`, `hard_target_updates`, `prep_training`, `prep_rollout`, same `train_info` keys. All
arithmetic of `train_policy_on_batch` (qmix.py:77-200) is two C-ABI calls:

    ope_qmix_loss_and_grad   forward of live+target nets, mixer, TD loss, full backward -> flat gradient
    ope_adam_step            clip_grad_norm_ + Adam (+ optional fused Polyak)

Live parameters, target parameters, Adam moments and gradients are flat CUDA vectors; `policy.q_network`,
`trainer.mixer` and their target twins are nn.Module views onto them (FlatModule), so checkpointing and rollout
see the same memory the kernels update. With torch.distributed initialised (one process per GPU) the flat
gradient is summed across ranks with ONE all-reduce between the two calls (off-policy_amd/dist.py).
"""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ... import dist as opdist
from ...config import require_reference_architecture
from ...utils.rec_buffer import StoreObs
from .algorithm.q_mixer import QMixer, VDNMixer


class FlatAdam(object):
    """Adam state over the trainer's flat parameter vector (torch.optim.Adam(lr, eps), qmix.py:71-72)."""

    def __init__(self, numel, lr, eps, device, betas=(0.9, 0.999), weight_decay=0.0):
        self.lr, self.eps, self.betas, self.weight_decay = lr, eps, betas, weight_decay
        self.exp_avg = torch.zeros(numel, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(numel, dtype=torch.float32, device=device)
        self.step_count = 0
        self.step_dev = None        # device int32 step counter (HIP-graph replays); None = host-side `step_count`

    def zero_grad(self):
        pass

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.lr, "eps": self.eps, "betas": self.betas, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class QMix(object):
    _mlp = False      # M_QMix overrides: non-recurrent agent nets on single transitions

    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=torch.device("cuda:0"), episode_length=None,
                 vdn=False):
        self.args = args
        # (the policies check their own support); one-layer hyper-networks: recurrent nets, one shared policy (checked below)
        # (VDN has no hyper-networks: args.hypernet_layers is irrelevant there and is ignored -- ADVICE r4)
        require_reference_architecture(args, allow_prev_act_inp=not self._mlp, allow_hypernet_layers_1=vdn or not self._mlp,
                                       allow_layer_N_2=not self._mlp, allow_no_feature_norm=True, allow_tanh=True)
        self.layer_N = int(getattr(args, "layer_N", 1))
        self.dims_flags = (0 if getattr(args, "use_feature_normalization", True) else _lib.OPE_DIMS_NO_FEATURE_NORM) | \
                          (0 if getattr(args, "use_ReLU", True) else _lib.OPE_DIMS_TANH)
        self.hypernet_layers = int(getattr(args, "hypernet_layers", 2)) if not vdn else 2
        self.use_popart = getattr(args, "use_popart", False)
        self.use_value_active_masks = getattr(args, "use_value_active_masks", False)
        self.use_per = args.use_per
        self.per_eps = args.per_eps
        self.use_huber_loss = args.use_huber_loss
        self.huber_delta = args.huber_delta
        self.device = torch.device(device)
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.lr, self.tau, self.opti_eps, self.weight_decay = args.lr, args.tau, args.opti_eps, args.weight_decay
        self.episode_length = args.episode_length if episode_length is None else episode_length
        self.num_agents = num_agents
        self.policies = policies
        self.policy_mapping_fn = policy_mapping_fn
        self.policy_ids = sorted(list(self.policies.keys()))
        self.policy_agents = {pid: sorted([a for a in range(num_agents) if policy_mapping_fn(a) == pid])
                              for pid in self.policies.keys()}
        self.use_same_share_obs = args.use_same_share_obs
        self.vdn = bool(vdn)
        self.multi = self.policy_ids != ["policy_0"] or len(self.policy_agents["policy_0"]) != num_agents
        self.fuse_soft_update = False       # True: Polyak inside ope_adam_step; soft_target_updates() then skips once
        self._polyak_done = False
        # per-trainer kernel choices / diagnostics, sent with every call in ope_qmix_cfg (0 = the library's choice by shape): two
        # trainers in one process do not share them. `mixer_path`: 1 resident-weight mixer, 2 streamed, 3 wide-state GEMM;
        # `time_chunks`: two-stream schedule; `scan_family` / `scan_waves`: GRU scan kernels; `debug`: keep intermediates.
        # `trunk_path`: 3 = one trunk launch per net (weights in registers), 4 = both nets in one launch (weights in LDS).
        # `chain_path`: 1 = head / mixer / TD / adjoints as four launches, 2 = the fused pair mixer_hyp + qchain (ope_chain.hip).
        # `live_rows`: 1 = every padded row of the batch is computed (as the reference does), 2 = only the rows before each episode's
        # termination (every other row is multiplied by a zero mask in the loss, qmix.py:161-166; found on the device from the sampled
        # dones_env every step), 0 = the library's choice by shape (ope_qmix_cfg.live_rows).
        self.tune = dict(mixer_path=0, time_chunks=0, scan_family=0, scan_waves=0, debug=0, trunk_path=0, chain_path=0, wgrad_path=0, live_rows=0)
        self._ws = {}
        self._ws_multi = {}
        self._gsq = {}
        # live-row plans built ahead of the step (RecPolicyBuffer.sample_inds(live_for=trainer); ope.h: ope_store_live_plan)
        self._live_seq = 0          # ticket of the latest plan
        self._live = None           # cached ope_live_target structs per (batch, workspace)
        if self.multi and (self.hypernet_layers == 1 or self.layer_N != 1 or self.dims_flags):
            raise NotImplementedError("hypernet_layers=1 / layer_N=2 / use_feature_normalization=False with several policies is not on the accelerated path")
        if (self.dims_flags & _lib.OPE_DIMS_NO_FEATURE_NORM) and float(getattr(args, "weight_decay", 0.0) or 0.0) != 0.0:
            raise NotImplementedError("use_feature_normalization=False with weight_decay: the constant feature_norm slots of the flat vector would decay")
        self._md_heads = None
        if self.multi:
            if any(getattr(p, "multidiscrete", False) for p in self.policies.values()):
                raise NotImplementedError("several policies under one mixer with a MultiDiscrete action space are not on the accelerated path")
            self._init_multi()
            if args.use_double_q:
                print("double Q learning will be used")
            return
        policy = self.policies["policy_0"]
        # MultiDiscrete action space (qmix.py:49-57, QMixPolicy.py:76-93): one q head per sub-action, one mixer input per (agent, sub-action).
        # The kernels know one head and one q value per agent, so every (agent, sub-action) pair is presented to them as an agent of its
        # own -- same observation rows, the stacked head of sum(act_dim) outputs, and an availability mask that is 1 on the sub-action's
        # block only (upstream passes no masks for these spaces), which restricts the greedy / target choices to that block; the chosen
        # action's q comes from the block's part of the stored one-hot row (`_md_expand`). Parameter gradients are sums over rows, so the
        # shared trunk / GRU, evaluated once per pair, gets exactly the sum of the heads' contributions. Costs n_heads x the agent-network
        # work: these spaces are MPE-sized.
        self._md_heads = list(policy.head_dims) if getattr(policy, "multidiscrete", False) else None
        if self._md_heads is not None:
            if self.vdn:
                raise NotImplementedError("VDN with a MultiDiscrete action space (upstream's VDNMixer fails on the recurrent trainer's input there too)")
            self.dims_flags |= _lib.OPE_DIMS_MASK_TARGET_MAX      # plain (non double-Q) targets: the per-head maximum
        self._n_kernel_agents = num_agents * (len(self._md_heads) if self._md_heads is not None else 1)
        self.num_mixer_q_inps = self._n_kernel_agents
        # the kernels see the network's input width: observation (+ previous one-hot action with prev_act_inp)
        self._dims = _lib.Dims(self._n_kernel_agents, policy.output_dim, policy.q_network_input_dim, policy.central_obs_dim, self.episode_length, self.layer_N, self.dims_flags)

        # ---- flat vectors: [agent | mixer], padded per tensor to 4 floats -------------------------------
        cfg = self._cfg(1)
        off = (C.c_int64 * 48)()
        siz = (C.c_int64 * 48)()
        P = _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz)
        if P < 0:
            _lib.check(int(P), "ope_qmix_param_layout")
        self.numel = int(P)
        self.theta = torch.zeros(self.numel, **self.tpdv)
        n_agent_tensors = _lib.OPE_QMIX_NPARAM_AGENT_MLP if self._mlp else (_lib.OPE_QMIX_NPARAM_AGENT_2 if self.layer_N == 2 else _lib.OPE_QMIX_NPARAM_AGENT)
        n_mixer_tensors = _lib.OPE_QMIX_NPARAM_MIXER_1 if self.hypernet_layers == 1 else _lib.OPE_QMIX_NPARAM_MIXER
        agent_numel = policy.q_network.padded_numel
        self.theta[:agent_numel].copy_(policy.q_network._flat[:agent_numel])
        policy.q_network.rebind(self.theta)            # live weights now live inside the trainer's flat vector
        if self.vdn:
            self.mixer = VDNMixer(args, num_agents, policy.central_obs_dim, self.device)
        else:
            self.mixer = QMixer(args, self._n_kernel_agents, policy.central_obs_dim, self.device, self.theta,
                                list(off)[n_agent_tensors:n_agent_tensors + n_mixer_tensors])
        # target networks: deep copies at construction (qmix.py:63-64)
        self.theta_tgt = self.theta.clone()
        self.target_policies = {"policy_0": _TargetPolicy(policy, policy.q_network.twin(self.theta_tgt))}
        if self.vdn:
            self.target_mixer = VDNMixer(args, num_agents, policy.central_obs_dim, self.device)
        else:
            self.target_mixer = QMixer(args, self._n_kernel_agents, policy.central_obs_dim, self.device, self.theta_tgt,
                                       list(off)[n_agent_tensors:n_agent_tensors + n_mixer_tensors], init=False)
        self.parameters = list(policy.parameters()) + list(self.mixer.parameters())
        self.optimizer = FlatAdam(self.numel, self.lr, self.opti_eps, self.device)
        self.grad = torch.zeros(self.numel + _lib.OPE_GRAD_TAIL, **self.tpdv)
        self._scratch = torch.zeros(int(_lib.lib.ope_adam_scratch_floats(self.numel)), **self.tpdv)
        self._stats = torch.zeros(4, **self.tpdv)
        if args.use_double_q:
            print("double Q learning will be used")

    # ---- several policies under one mixer (share_policy = False; qmix.py:100-150, mqmix.py:95-178) ----------------
    def _init_multi(self):
        """Flat vectors [agent_0 | agent_1 | ... | mixer] in `policy_ids` order (one Adam over everything, one clip norm:
        qmix.py:66-72,190-193). Policies may differ in observation width, action count and number of agents; their agents must be
        numbered policy by policy (the order the reference concatenates the agents' q values in, qmix.py:150)."""
        args, T = self.args, self.episode_length
        flat_agents = [a for p in self.policy_ids for a in self.policy_agents[p]]
        if flat_agents != list(range(self.num_agents)) or any(len(self.policy_agents[p]) == 0 for p in self.policy_ids):
            raise NotImplementedError("several policies: agents must be numbered policy by policy, every policy with at least one agent")
        first = self.policies[self.policy_ids[0]]
        S = first.central_obs_dim
        self._pdims, self._poff, self._a0 = {}, {}, {}
        off, a0 = 0, 0
        for p in self.policy_ids:
            pol = self.policies[p]
            self._pdims[p] = _lib.Dims(len(self.policy_agents[p]), pol.act_dim, pol.q_network_input_dim, S, T)
            self._poff[p], self._a0[p] = off, a0
            off += pol.q_network.padded_numel
            a0 += len(self.policy_agents[p])
        self._dims = _lib.Dims(self.num_agents, first.act_dim, first.q_network_input_dim, S, T)      # the mixing part's cfg
        self._mixer_off = off
        self._shift = off - first.q_network.padded_numel      # the joint call's theta / grad start here (ope.h, ope_qmix_cfg.phase)
        offs = (C.c_int64 * 48)()
        P = _lib.lib.ope_qmix_param_layout(C.byref(self._cfg(1)), offs, None)
        if P < 0:
            _lib.check(int(P), "ope_qmix_param_layout")
        self.numel = self._shift + int(P)
        self.theta = torch.zeros(self.numel, **self.tpdv)
        for p in self.policy_ids:
            q = self.policies[p].q_network
            o, n = self._poff[p], q.padded_numel
            self.theta[o:o + n].copy_(q._flat[:n])
            q.rebind(self.theta[o:o + n])
        n_agent_tensors = _lib.OPE_QMIX_NPARAM_AGENT_MLP if self._mlp else _lib.OPE_QMIX_NPARAM_AGENT
        moffs = [self._shift + int(x) for x in list(offs)[n_agent_tensors:n_agent_tensors + _lib.OPE_QMIX_NPARAM_MIXER]]
        if self.vdn:
            self.mixer = VDNMixer(args, self.num_agents, S, self.device)
        else:
            self.mixer = QMixer(args, self.num_agents, S, self.device, self.theta, moffs)
        self.theta_tgt = self.theta.clone()
        self.target_policies = {}
        for p in self.policy_ids:
            q = self.policies[p].q_network
            o, n = self._poff[p], q.padded_numel
            self.target_policies[p] = _TargetPolicy(self.policies[p], q.twin(self.theta_tgt[o:o + n]))
        if self.vdn:
            self.target_mixer = VDNMixer(args, self.num_agents, S, self.device)
        else:
            self.target_mixer = QMixer(args, self.num_agents, S, self.device, self.theta_tgt, moffs, init=False)
        self.parameters = [x for p in self.policies.values() for x in p.parameters()] + list(self.mixer.parameters())
        self.optimizer = FlatAdam(self.numel, self.lr, self.opti_eps, self.device)
        self.grad = torch.zeros(self.numel + _lib.OPE_GRAD_TAIL, **self.tpdv)
        self._scratch = torch.zeros(int(_lib.lib.ope_adam_scratch_floats(self.numel)), **self.tpdv)
        self._stats = torch.zeros(4, **self.tpdv)

    def _part_cfg(self, pid, batch, phase):
        cfg = self._cfg(batch)
        cfg.dims, cfg.vdn, cfg.phase = self._pdims[pid], 1, phase
        return cfg

    def _ws_for(self, key, cfg):
        # (its own cache, keyed by everything the workspace plan depends on: `_workspace` below replaces its dict when the plan changes)
        key = key + (cfg.mixer_path, cfg.time_chunks, cfg.chain_path)
        if key not in self._ws_multi:
            need = _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg))
            if need < 0:
                _lib.check(int(need), "ope_qmix_workspace_bytes")
            ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib.ope_qmix_workspace_init(C.byref(cfg), _lib.ptr(ws), ws.numel(), _lib.current_stream()), "ope_qmix_workspace_init")
            views = {}
            for name in ("agent_q", "agent_nq", "d_agent_q"):
                n = C.c_int64(0)
                o = _lib.lib.ope_qmix_workspace_find(C.byref(cfg), name.encode(), C.byref(n))
                views[name] = ws[o:o + 4 * n.value].view(torch.float32).view(cfg.dims.episode_length, cfg.batch, cfg.dims.n_agents)
            self._ws_multi[key] = (ws, views)
        return self._ws_multi[key]

    def _train_multi(self, parts, share, rew, dones_env, importance_weights, idxes):
        """One update with several policies. parts[pid] = (obs [T+1, n_p, B, D_p], acts [T, n_p, B, A_p], avail or None) in the
        kernels' layout; share [T+1, B, S]; rew [T, 1, B, 1] (the agents share it); dones_env [T, B, 1]. Three kinds of C-ABI calls
        (ope_qmix_cfg.phase): every policy's networks forward -> the mixer / VDN sum, TD loss and mixer gradients over all agents'
        q values -> every policy's networks backward; then ONE all-reduce (data parallel) and ONE clip + Adam over the whole vector."""
        T, N = self.episode_length, self.num_agents
        B = int(share.shape[1])
        st = _lib.current_stream()
        jcfg = self._cfg(B)
        jcfg.phase = 2
        jws, jv = self._ws_for(("joint", B), jcfg)
        keep = []
        calls = {}
        for p in self.policy_ids:
            obs, acts, avail = parts[p]
            assert obs.shape[0] == T + 1 and obs.shape[1] == self._pdims[p].n_agents and obs.shape[2] == B, "batch does not match the trainer's dimensions"
            cfg = self._part_cfg(p, B, 1)
            ws, v = self._ws_for((p, B), cfg)
            f = _lib.Fields()
            f.obs, f.acts, f.avail_acts = _lib.ptr(obs).value, _lib.ptr(acts).value, _lib.ptr(avail).value
            o, n = self._poff[p], self.policies[p].q_network.padded_numel
            th, tht, gr = self.theta[o:o + n], self.theta_tgt[o:o + n], self.grad[o:o + n]
            calls[p] = (cfg, f, th, tht, gr, ws, v)
            _lib.check(_lib.lib.ope_qmix_loss_and_grad(C.byref(cfg), C.byref(f), _lib.ptr(th), _lib.ptr(tht), None, _lib.ptr(ws), ws.numel(),
                                                       _lib.ptr(gr), None, st), "ope_qmix_loss_and_grad(phase 1)")
            a0, n_p = self._a0[p], self._pdims[p].n_agents
            jv["agent_q"][:, :, a0:a0 + n_p].copy_(v["agent_q"])
            jv["agent_nq"][:, :, a0:a0 + n_p].copy_(v["agent_nq"])
            keep.append((obs, acts, avail))
        w = td_stats = None
        if self.use_per:
            w = (importance_weights.to(self.device, dtype=torch.float32).contiguous() if torch.is_tensor(importance_weights) else
                 torch.as_tensor(np.asarray(importance_weights), dtype=torch.float32).to(self.device).contiguous())
            td_stats = torch.empty(2 * B, **self.tpdv)
        rew_all = rew.expand(T, N, B, 1).contiguous()          # the TD kernel reads agent 0's slab of [T][N][B][1]
        f = _lib.Fields()
        f.share_obs, f.rewards, f.dones_env = _lib.ptr(share).value, _lib.ptr(rew_all).value, _lib.ptr(dones_env).value
        th, tht, gr = self.theta[self._shift:], self.theta_tgt[self._shift:], self.grad[self._shift:]
        _lib.check(_lib.lib.ope_qmix_loss_and_grad(C.byref(jcfg), C.byref(f), _lib.ptr(th), _lib.ptr(tht), _lib.ptr(w), _lib.ptr(jws), jws.numel(),
                                                   _lib.ptr(gr), _lib.ptr(td_stats), st), "ope_qmix_loss_and_grad(phase 2)")
        for p in self.policy_ids:
            cfg, f, th, tht, gr, ws, v = calls[p]
            a0, n_p = self._a0[p], self._pdims[p].n_agents
            v["d_agent_q"].copy_(jv["d_agent_q"][:, :, a0:a0 + n_p])
            cfg.phase = 3
            _lib.check(_lib.lib.ope_qmix_loss_and_grad(C.byref(cfg), C.byref(f), _lib.ptr(th), _lib.ptr(tht), None, _lib.ptr(ws), ws.numel(),
                                                       _lib.ptr(gr), None, st), "ope_qmix_loss_and_grad(phase 3)")
        _, world_size = opdist.world()
        opdist.allreduce_flat_(self.grad)
        self.optimizer.step_count += 1
        ac = _lib.AdamCfg()
        ac.lr, ac.beta1, ac.beta2, ac.eps = self.lr, self.optimizer.betas[0], self.optimizer.betas[1], self.opti_eps
        ac.max_grad_norm, ac.weight_decay = float(self.args.max_grad_norm), 0.0
        ac.tau, ac.do_polyak = float(self.tau), int(self.fuse_soft_update)
        ac.step = self.optimizer.step_count
        ac.qtot_denominator = float(T * B * world_size)
        stats = torch.empty(4, **self.tpdv)
        _lib.check(_lib.lib.ope_adam_step(C.byref(ac), self.numel, _lib.ptr(self.theta), _lib.ptr(self.theta_tgt),
                                          _lib.ptr(self.optimizer.exp_avg), _lib.ptr(self.optimizer.exp_avg_sq),
                                          _lib.ptr(self.grad), _lib.ptr(self._scratch), _lib.ptr(stats), st), "ope_adam_step")
        self._polyak_done = bool(self.fuse_soft_update)
        train_info = {"loss": stats[0], "grad_norm": stats[1], "Q_tot": stats[2]}
        new_priorities = None
        if self.use_per and torch.is_tensor(importance_weights):
            s = td_stats.view(B, 2)
            new_priorities = ((1 - self.args.per_nu) * s[:, 0] + self.args.per_nu * s[:, 1]) + self.per_eps
        elif self.use_per:
            s = td_stats.view(B, 2).cpu().numpy().astype(np.float32)
            new_priorities = ((1 - self.args.per_nu) * s[:, 0] + self.args.per_nu * s[:, 1]).flatten() + self.per_eps
        self._last = (keep, share, rew_all, dones_env, w, td_stats)    # keep inputs alive past the async launches
        return train_info, new_priorities, idxes

    # ---- helpers --------------------------------------------------------------------------------------------
    def _cfg(self, batch):
        a = self.args
        cfg = _lib.QmixCfg()
        cfg.dims = self._dims
        cfg.batch = int(batch)
        cfg.vdn = int(self.vdn)
        cfg.use_double_q = int(bool(a.use_double_q))
        cfg.use_huber = int(bool(a.use_huber_loss))
        cfg.use_per = int(bool(a.use_per))
        cfg.gamma, cfg.huber_delta = float(a.gamma), float(a.huber_delta)
        cfg.per_nu, cfg.per_eps = float(a.per_nu), float(a.per_eps)
        cfg.mlp = int(self._mlp)
        t = self.tune
        cfg.mixer_path, cfg.time_chunks = int(t["mixer_path"]), int(t["time_chunks"])
        cfg.scan_family, cfg.scan_waves, cfg.debug = int(t["scan_family"]), int(t["scan_waves"]), int(t["debug"])
        cfg.trunk_path = int(t["trunk_path"])
        cfg.chain_path = int(t.get("chain_path", 0))
        cfg.wgrad_path = int(t.get("wgrad_path", 0))      # 1 = one tile per wave (wgrad), 2 = register-blocked (wgrad2), 0 = by shape
        cfg.live_rows = int(t.get("live_rows", 0))
        cfg.hypernet_layers = 0 if self.vdn else int(getattr(self, "hypernet_layers", 2))
        return cfg

    def _workspace(self, cfg):
        B = cfg.batch
        shape_key = (cfg.mixer_path, cfg.time_chunks, cfg.chain_path)       # the workspace plan depends on these
        if self._ws.get("_key") != shape_key:
            self._ws = {"_key": shape_key}
            self._gsq = {}
        if B not in self._ws:
            need = _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg))
            if need < 0:
                _lib.check(int(need), "ope_qmix_workspace_bytes")
            ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib.ope_qmix_workspace_init(C.byref(cfg), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                       "ope_qmix_workspace_init")
            self._ws[B] = ws
        return self._ws[B]

    def build_live_plan(self, pbuf, host_inds, batch, live_only=False, region=0):
        """Have the NEXT gather launch of `pbuf` (RecPolicyBuffer; the call that follows in sample_inds) also build the live-row plan of this
        trainer's next step on those `batch` episodes -- a few extra workgroups in front of the copy's that read the store's termination
        flags through the launch's own indices (ope.h: ope_store_gather_attach_live) -- and return the tag `train_policy_on_batch`
        recognises; None where that step would not run on live rows "by shape" (or the trainer pins every padded row / is one of the
        multi-policy, MLP forms). Same stream as the step: no events, one plan region. `live_only`: that launch's copy also leaves the time
        entries at and behind each episode's length unwritten in obs / share_obs (ope_live_target.copy_live_only; tag[4]). `region` (0 | 1):
        which of the workspace's two plan regions (ope_qmix_live_target) -- a batch gathered AHEAD, while the step before still reads its own
        plan, goes to the other one (RecPolicyBuffer.sample_inds_ahead alternates them); a tag stays good until the next plan for its region."""
        if self.multi or self._mlp or int(self.tune.get("live_rows", 0)) == 1 or int(self.tune.get("debug", 0)):
            return None
        batch = int(batch)
        cfg = self._cfg(batch)
        if int(pbuf.episode_length) != int(self.episode_length) or not _lib.lib.ope_qmix_live_rows_ok(C.byref(cfg)):
            return None
        ws = self._workspace(cfg)
        if self._live is None:
            self._live = {"targets": {}}
        region = int(region) & 1
        key = (batch, int(ws.data_ptr()), region)
        tgt = self._live["targets"].get(key)
        if tgt is None:
            tgt = _lib.LiveTarget()
            if _lib.lib.ope_qmix_live_target(C.byref(cfg), _lib.ptr(ws), ws.numel(), region, C.byref(tgt)) != 0:
                return None
            self._live["targets"][key] = tgt
        tgt.copy_live_only = 1 if live_only else 0
        _lib.check(_lib.lib.ope_store_gather_attach_live(C.byref(tgt)), "ope_store_gather_attach_live")
        self._live_seq += 1
        self._live.setdefault("latest", {})[(int(ws.data_ptr()), region)] = self._live_seq
        return ("ope_live", int(ws.data_ptr()), self._live_seq, region, bool(live_only))

    def workspace_view(self, batch, name):
        """Debug/test access to a named intermediate of the last step with this batch size (float32 view)."""
        cfg = self._cfg(batch)
        n = C.c_int64(0)
        off = _lib.lib.ope_qmix_workspace_find(C.byref(cfg), name.encode(), C.byref(n))
        if off < 0:
            raise KeyError(name)
        return self._ws[batch][off:off + 4 * n.value].view(torch.float32)

    def _to_device_layout(self, x, agent_axis):
        """Bring one batch field to the kernels' layout: [T(+1), N, B, dim] (agent fields) or [T(+1), B, dim]."""
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

    # ---- the training step ------------------------------------------------------------------------------------
    def train_policy_on_batch(self, batch, update_policy_id=None):
        """See offpolicy/algorithms/qmix/qmix.py:77-200. `batch` is the 9-tuple from `buffer.sample()` (CUDA tensors
        from our buffer, or numpy arrays from the reference's). Returns (train_info, new_priorities, idxes)."""
        obs_b, cent_b, act_b, rew_b, dones_b, dones_env_b, avail_b, importance_weights, idxes = batch
        if self.multi:
            return self._train_multi_rec(batch)
        pid = self.policy_ids[0]
        obs = obs_b[pid]
        if isinstance(obs, StoreObs):
            # observations left in the replay store (RecPolicyBuffer.lazy_obs): the first-layer kernels read the rows in place where the
            # configuration allows it (ope_qmix_obs_ref_ok); otherwise they are gathered now, as sample_inds would have
            if getattr(self.policies[pid], "prev_act_inp", False) or not self.obs_ref_ok(obs.batch):
                obs = obs.materialize()
        if not isinstance(obs, StoreObs):
            obs = self._to_device_layout(obs, True)
        # the mixer's state: the shared centralized observation, or agent 0's when every agent has its own (qmix.py:86-90)
        cent = cent_b[pid] if self.use_same_share_obs else cent_b[pid][0]
        share = self._to_device_layout(cent, False)
        acts = self._to_device_layout(act_b[pid], True)
        rew = self._to_device_layout(rew_b[pid], True)
        dones_env = self._to_device_layout(dones_env_b[pid], False)
        live_tag = getattr(dones_env_b[pid], "_ope_live", None)      # the gather that wrote this batch built the step's row plan (sample_inds(live_for=))
        avail = self._to_device_layout(avail_b[pid], True) if (avail_b is not None and avail_b[pid] is not None) else None
        if getattr(self.policies[pid], "prev_act_inp", False):
            # qmix.py:123-124: the network input at step t is [obs_t | action taken at t-1], zeros at t = 0
            prev = torch.cat((torch.zeros_like(acts[:1]), acts), dim=0)
            obs = torch.cat((obs, prev), dim=-1).contiguous()
        if self._md_heads is not None:
            obs, acts, rew, avail = self._md_expand(obs, acts, rew, avail)
        return self._train_on_device_batch(obs, share, acts, rew, dones_env, avail, importance_weights, idxes, live_tag=live_tag)

    def _md_expand(self, obs, acts, rew, avail):
        """MultiDiscrete: the batch as the kernels see it -- one "agent" per (agent, sub-action), agent-major (the order upstream concatenates
        the heads' q values in, QMixPolicy.py:76-93 + qmix.py:134-136). obs / rewards rows repeated per head; the stored action row keeps the
        head's one-hot block only (its first maximum is then the sub-action taken); the availability mask is the head's block."""
        assert avail is None, "MultiDiscrete action spaces come without availability masks"
        if isinstance(obs, StoreObs):
            obs = self._to_device_layout(obs.materialize(), True)
        heads = self._md_heads
        Hn, A = len(heads), int(sum(heads))
        T1, N, B, _ = obs.shape
        block = torch.zeros(Hn, A, **self.tpdv)
        lo = 0
        for h, d in enumerate(heads):
            block[h, lo:lo + d] = 1.0
            lo += d
        mask = block.repeat(N, 1)[None, :, None, :]                                  # [1, N * Hn, 1, A]: row p = agent * Hn + head
        obs_x = obs.repeat_interleave(Hn, dim=1).contiguous()
        acts_x = (acts.repeat_interleave(Hn, dim=1) * mask).contiguous()
        rew_x = rew.repeat_interleave(Hn, dim=1).contiguous()
        avail_x = mask.expand(T1, N * Hn, B, A).contiguous()
        return obs_x, acts_x, rew_x, avail_x

    def _train_multi_rec(self, batch):
        obs_b, cent_b, act_b, rew_b, dones_b, dones_env_b, avail_b, importance_weights, idxes = batch
        first, last = self.policy_ids[0], self.policy_ids[-1]
        parts = {}
        for pid in self.policy_ids:
            obs = self._to_device_layout(obs_b[pid], True)
            acts = self._to_device_layout(act_b[pid], True)
            avail = self._to_device_layout(avail_b[pid], True) if (avail_b is not None and avail_b[pid] is not None) else None
            if getattr(self.policies[pid], "prev_act_inp", False):
                prev = torch.cat((torch.zeros_like(acts[:1]), acts), dim=0)
                obs = torch.cat((obs, prev), dim=-1).contiguous()
            parts[pid] = (obs, acts, avail)
        cent = cent_b[first] if self.use_same_share_obs else cent_b[first][0]            # qmix.py:86-90: the first policy's
        share = self._to_device_layout(cent, False)
        dones_env = self._to_device_layout(dones_env_b[first], False)
        rew = self._to_device_layout(rew_b[last], True)[:, :1]                               # qmix.py:103,159: the LAST policy's agent 0
        return self._train_multi(parts, share, rew, dones_env, importance_weights, idxes)

    def obs_ref_ok(self, batch):
        """Can a step on `batch` episodes read its observation rows from the replay store (StoreObs) instead of a gathered tensor?"""
        if self.multi or self._mlp or getattr(self, "_md_heads", None) is not None:
            return False
        return bool(_lib.lib.ope_qmix_obs_ref_ok(C.byref(self._cfg(int(batch)))))

    def _train_on_device_batch(self, obs, share, acts, rew, dones_env, avail, importance_weights, idxes, live_tag=None):
        oref = None
        if isinstance(obs, StoreObs):
            N, T1, B, D = obs.shape
            oref = obs.ref()
        else:
            T1, N, B, D = obs.shape
        assert T1 == self.episode_length + 1 and N == getattr(self, "_n_kernel_agents", self.num_agents), "batch does not match the trainer's dimensions"
        cfg = self._cfg(B)
        ws = self._workspace(cfg)
        if (live_tag is not None and oref is None and cfg.live_rows in (0, 2) and self._live is not None and live_tag[0] == "ope_live" and
                live_tag[1] == int(ws.data_ptr()) and self._live.get("latest", {}).get((live_tag[1], int(live_tag[3]))) == live_tag[2]):
            # the gather launch that wrote this batch -- the latest one attached for this workspace -- built the step's plan: no plan launch
            cfg.live_rows = 3 + int(live_tag[3])
        elif live_tag is not None and len(live_tag) > 4 and live_tag[4]:
            # the gather left the post-terminal entries of obs / share_obs unwritten for the live-row step of ITS plan: no other step may read them
            raise RuntimeError("this batch was gathered with live_only=True: only the live-row step it was sampled for (the trainer's next "
                               "train_policy_on_batch, same batch size and tuning) can consume it")
        f = _lib.Fields()
        f.obs = None if oref is not None else _lib.ptr(obs).value
        f.share_obs, f.acts, f.rewards = _lib.ptr(share).value, _lib.ptr(acts).value, _lib.ptr(rew).value
        f.dones, f.dones_env, f.avail_acts = None, _lib.ptr(dones_env).value, _lib.ptr(avail).value
        w = None
        td_stats = None
        if self.use_per:
            w = (importance_weights.to(self.device, dtype=torch.float32).contiguous() if torch.is_tensor(importance_weights) else
                 torch.as_tensor(np.asarray(importance_weights), dtype=torch.float32).to(self.device).contiguous())
            td_stats = torch.empty(2 * B, **self.tpdv)
        st = _lib.current_stream()
        _, world_size = opdist.world()
        n_head = self.numel + _lib.OPE_GRAD_TAIL
        dev_prio = self.use_per and torch.is_tensor(importance_weights)
        if dev_prio and world_size > 1 and self.grad.numel() < n_head + B * world_size:
            # room behind the tail for the ranks' per-episode priorities: they ride on the gradient all-reduce (dist.priority_slots)
            self.grad = torch.zeros(n_head + B * world_size, **self.tpdv)
        if oref is not None:
            _lib.check(_lib.lib.ope_qmix_loss_and_grad_ref(C.byref(cfg), C.byref(f), C.byref(oref), _lib.ptr(self.theta), _lib.ptr(self.theta_tgt),
                                                           _lib.ptr(w), _lib.ptr(ws), ws.numel(), _lib.ptr(self.grad),
                                                           _lib.ptr(td_stats), st), "ope_qmix_loss_and_grad_ref")
        else:
            _lib.check(_lib.lib.ope_qmix_loss_and_grad(C.byref(cfg), C.byref(f), _lib.ptr(self.theta), _lib.ptr(self.theta_tgt),
                                                       _lib.ptr(w), _lib.ptr(ws), ws.numel(), _lib.ptr(self.grad),
                                                       _lib.ptr(td_stats), st), "ope_qmix_loss_and_grad")
        gathered = None
        if dev_prio and world_size > 1:
            s = td_stats.view(B, 2)
            gathered = opdist.priority_slots(self.grad, n_head, ((1 - self.args.per_nu) * s[:, 0] + self.args.per_nu * s[:, 1]) + self.per_eps)
            opdist.allreduce_flat_(self.grad[:n_head + B * world_size])
        else:
            opdist.allreduce_flat_(self.grad[:n_head])           # no-op on one GPU; ONE collective otherwise
        self.optimizer.step_count += 1
        ac = _lib.AdamCfg()
        ac.lr, ac.beta1, ac.beta2, ac.eps = self.lr, self.optimizer.betas[0], self.optimizer.betas[1], self.opti_eps
        ac.max_grad_norm, ac.weight_decay = float(self.args.max_grad_norm), 0.0   # QMix's Adam ignores weight_decay (A-8)
        ac.tau, ac.do_polyak = float(self.tau), int(self.fuse_soft_update)
        ac.step = self.optimizer.step_count
        if self.optimizer.step_dev is not None:     # HIP-graph replays: the count advances on the device
            ac.step_counter = _lib.ptr(self.optimizer.step_dev).value
        ac.qtot_denominator = float(self.episode_length * B * world_size)
        if world_size == 1:
            # the finalize launch of ope_qmix_loss_and_grad left per-workgroup partial sums of grad^2 in the workspace: the
            # optimizer call then needs no norm pass of its own (not valid for an all-reduced gradient)
            gsq = self._gsq.get(B)
            if gsq is None:
                n = C.c_int64(0)
                off = _lib.lib.ope_qmix_workspace_find(C.byref(cfg), b"gsq_part", C.byref(n))
                gsq = self._gsq[B] = (int(off), int(n.value)) if off >= 0 else (-1, 0)
            if gsq[0] >= 0:
                ac.sumsq_partials = ws.data_ptr() + gsq[0]
                ac.n_sumsq_partials = gsq[1]
        stats = torch.empty(4, **self.tpdv)
        _lib.check(_lib.lib.ope_adam_step(C.byref(ac), self.numel, _lib.ptr(self.theta), _lib.ptr(self.theta_tgt),
                                          _lib.ptr(self.optimizer.exp_avg), _lib.ptr(self.optimizer.exp_avg_sq),
                                          _lib.ptr(self.grad), _lib.ptr(self._scratch), _lib.ptr(stats), st), "ope_adam_step")
        self._polyak_done = bool(self.fuse_soft_update)
        train_info = {"loss": stats[0], "grad_norm": stats[1], "Q_tot": stats[2]}
        new_priorities = None
        # Uniform contract (dist.allgather_cat): the LOCAL share's priorities are returned, whatever the world size; what the all-reduce
        # already gathered is kept beside them (a view into the gradient vector, valid until the next step) for callers that want to
        # skip the extra collective: dist.allgather_cat(new_priorities, have=trainer.gathered_priorities)
        self.gathered_priorities = gathered
        if gathered is not None:
            rank, _ = opdist.world()
            new_priorities = gathered[rank * B:(rank + 1) * B].clone()
        elif self.use_per and torch.is_tensor(importance_weights):   # device trees (device_tree=True): priorities stay in HBM
            s = td_stats.view(B, 2)
            new_priorities = ((1 - self.args.per_nu) * s[:, 0] + self.args.per_nu * s[:, 1]) + self.per_eps
        elif self.use_per:
            s = td_stats.view(B, 2).cpu().numpy().astype(np.float32)
            new_priorities = ((1 - self.args.per_nu) * s[:, 0] + self.args.per_nu * s[:, 1]).flatten() + self.per_eps
        self._last = (obs, share, acts, rew, dones_env, avail, w, td_stats)   # keep inputs alive past the async launch
        return train_info, new_priorities, idxes

    def make_graphed_step(self, buffer, batch_size, policy_id="policy_0", gather_in_graph=True):
        """`buffer.sample` on given episode indices + `train_policy_on_batch` + `soft_target_updates`, captured ONCE as a HIP
        graph and replayed with one launch per step. The eager step is 17 kernels whose enqueueing costs the host about as
        long as they run (0.47 ms at 3s5z, B=32), so the GPU idles ~20 us per step waiting for the first launches; a replay
        removes the host from the loop. Returns `step(inds) -> train_info` (`inds`: numpy int64 [batch_size], e.g.
        np.random.choice(len(buffer), batch_size); train_info: device tensors overwritten by every replay).
        With gather_in_graph=False the gather stays an eager launch into a fixed batch (so it can be bracketed by timing
        events: `step(inds, timing_events=(start, end))`) and the graph holds the 16 training kernels.
        Restrictions: uniform replay (PER's importance weights come from the caller per step), Adam's step count lives on the device.
        In a multi-process run (one rank per GPU; `batch_size` and `inds` = the rank's share) the gradient all-reduce is captured with the
        kernels: that needs the one-shot xGMI exchange (dist.setup_fast_allreduce verified it; its call counter lives on the device) --
        with the RCCL fallback the step stays eager. The graph holds pointers into that exchange's buffers: it refuses to replay once
        dist.disable_fast_allreduce has retired it."""
        if self.use_per or self.multi or getattr(self, "_md_heads", None) is not None:
            raise NotImplementedError("graphed step: uniform replay, one shared policy, Discrete actions")
        if opdist.is_distributed() and not opdist.graph_safe_allreduce(self.numel + _lib.OPE_GRAD_TAIL):
            raise NotImplementedError("graphed step at world > 1 needs the one-shot all-reduce with slots that hold the gradient vector (%s)" % opdist.allreduce_backend())
        pbuf = buffer.policy_buffers[policy_id]
        B = int(batch_size)
        self.fuse_soft_update = True
        opt = self.optimizer
        opt.step_dev = torch.tensor([opt.step_count, 0], dtype=torch.int32, device=self.device)    # [count, ticket]
        static_inds = torch.zeros(B, dtype=torch.int64, device=self.device)

        static_batch = None if gather_in_graph else pbuf.alloc_batch(B)

        def train(s):
            info, _, _ = self.train_policy_on_batch(tuple({policy_id: x} for x in s) + (None, None))
            self.soft_target_updates()
            return info

        def body():
            if gather_in_graph:
                return train(pbuf.sample_inds(static_inds))
            return train(self._static_sample)
        # The warm-up below really trains (two optimizer + Polyak updates on a throw-away batch): snapshot the model and
        # optimizer state and put it back afterwards, so that building a graphed step leaves the trainer exactly where an
        # eager run would be. (Sticky by design: fuse_soft_update stays on -- later eager calls fold Polyak into Adam too.)
        snap = (self.theta.clone(), self.theta_tgt.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_dev.clone(),
                opt.step_count, self._polyak_done)
        host_rng = np.random.get_state()
        side = torch.cuda.Stream(device=self.device)     # warm-up off the capture: workspace, allocator pools, lazy init
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            static_inds.copy_(torch.from_numpy(np.random.choice(len(buffer), B)).to(self.device))
            if not gather_in_graph:
                self._static_sample = pbuf.sample_inds(static_inds, out=static_batch)
            for _ in range(2):
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        np.random.set_state(host_rng)                    # the throw-away index draw is not part of the caller's stream
        self.theta.copy_(snap[0]); self.theta_tgt.copy_(snap[1]); opt.exp_avg.copy_(snap[2]); opt.exp_avg_sq.copy_(snap[3])
        opt.step_dev.copy_(snap[4]); opt.step_count = snap[5]; self._polyak_done = snap[6]
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            info = body()
        opt.step_count = int(opt.step_dev[0].item())      # capture ran the host code but no kernels
        ring = [(torch.empty(B, dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        state = {"k": 0, "used": [False] * 8}
        ar_gen = opdist.note_graph_capture() if opdist.is_distributed() else None

        def step(inds, timing_events=None):
            if ar_gen is not None and opdist.fast_generation() != ar_gen:
                raise RuntimeError("this graphed step captured the one-shot all-reduce, which has since been disabled (%s): build a new "
                                   "graphed step or train eagerly" % opdist.allreduce_backend())
            k = state["k"]
            state["k"] = (k + 1) % 8
            host, ev = ring[k]
            if state["used"][k]:
                ev.synchronize()
            host.copy_(torch.from_numpy(np.asarray(inds, dtype=np.int64)))
            static_inds.copy_(host, non_blocking=True)
            ev.record()
            state["used"][k] = True
            if not gather_in_graph:
                pbuf.sample_inds(static_inds, timing_events=timing_events, out=static_batch)
            graph.replay()
            opt.step_count += 1
            return info
        self._graph = (graph, static_inds, ring, static_batch)      # keep alive
        return step

    # ---- target updates ---------------------------------------------------------------------------------------
    def hard_target_updates(self):
        print("hard update targets")
        _lib.check(_lib.lib.ope_polyak(self.numel, _lib.ptr(self.theta), _lib.ptr(self.theta_tgt), 1.0,
                                       _lib.current_stream()), "ope_polyak")

    def soft_target_updates(self):
        if self._polyak_done:            # already applied inside ope_adam_step for this step
            self._polyak_done = False
            return
        _lib.check(_lib.lib.ope_polyak(self.numel, _lib.ptr(self.theta), _lib.ptr(self.theta_tgt), float(self.tau),
                                       _lib.current_stream()), "ope_polyak")

    def prep_training(self):
        pass

    def prep_rollout(self):
        pass


class _TargetPolicy(object):
    """Target twin of a policy: same hyper-parameters, q_network bound to the target flat vector."""

    def __init__(self, policy, q_network):
        self.__dict__.update({k: v for k, v in policy.__dict__.items() if k != "q_network"})
        self.q_network = q_network
        self._cls = policy.__class__

    def __getattr__(self, name):
        fn = getattr(self._cls, name)
        return fn.__get__(self, self._cls)

    def parameters(self):
        return self.q_network.parameters()

    def load_state(self, source_policy):
        self.q_network.load_state_dict(source_policy.q_network.state_dict())
