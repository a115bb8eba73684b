"""Recurrent MADDPG / MATD3 policy: recurrent actor + centralised recurrent critic (+ targets, optimizer state).

Mirror of offpolicy/algorithms/r_maddpg/algorithm/rMADDPGPolicy.py:11-176 for discrete, multi-discrete and continuous action spaces. The
four networks are drawn in the reference's construction order (actor, critic, target actor, target critic) and the
targets then take the live weights, so equal seeds give equal weights.
"""
import numpy as np
import torch
from torch.distributions import OneHotCategorical

from .... import _lib
from ....config import require_reference_architecture
from ....utils.spaces import get_dim_from_space
from ...maddpg.algorithm.MADDPGPolicy import gumbel_softmax_hard, onehot_from_logits, sample_gumbel_uniform
from ...qmix.algorithm.QMixPolicy import DecayThenFlatSchedule
from ...qmix.qmix import FlatAdam
from .r_actor_critic import R_MADDPG_Actor, R_MADDPG_Critic, draw_ractor_values, draw_rcritic_values


class R_MADDPGPolicy(object):
    def __init__(self, config, policy_config, target_noise=None, td3=False, train=True):
        self.config = config
        self.device = torch.device(config["device"])
        self.args = config["args"]
        require_reference_architecture(self.args)
        a = self.args
        self.tau, self.lr, self.opti_eps, self.weight_decay = a.tau, a.lr, a.opti_eps, a.weight_decay
        self.prev_act_inp = bool(getattr(a, "prev_act_inp", False))
        if self.prev_act_inp:
            raise NotImplementedError("prev_act_inp=True is not on the accelerated path")
        self.central_obs_dim, self.central_act_dim = policy_config["cent_obs_dim"], policy_config["cent_act_dim"]
        self.obs_space, self.act_space = policy_config["obs_space"], policy_config["act_space"]
        self.obs_dim, self.act_dim = get_dim_from_space(self.obs_space), get_dim_from_space(self.act_space)
        self.hidden_size = a.hidden_size
        kind = self.act_space.__class__.__name__
        self.discrete, self.multidiscrete = kind != "Box", "MultiDiscrete" in kind      # util.py:271-281; Box = continuous (rMADDPGPolicy.py:121-129)
        # act_dim: as upstream an int, or the ARRAY of a multi-discrete space's sub-action sizes; output_dim: the action vector's length
        self.output_dim = int(sum(self.act_dim)) if self.multidiscrete else self.act_dim
        if self.multidiscrete and len(self.act_dim) > 6:
            raise NotImplementedError("multi-discrete action spaces with more than 6 sub-actions are not on the accelerated path")
        self.target_noise = target_noise
        self.td3 = bool(td3)
        self.num_q = 2 if td3 else 1
        # the joint action is this policy's width times the number of agents, unless policies of OTHER action dimensions share the critic
        # (share_policy = False on e.g. simple_speaker_listener): then only its total width is known here (ope_rddpg_cfg.joint_act_dim)
        # (`policy_config["num_agents"]`, optional, settles it; the guess can be wrong for policies of different widths that divide the total --
        # the trainer overwrites dims.n_agents / joint_act_dim with the real layout before any launch: MADDPGPolicy.py has the same note)
        self.mixed_act_dims = self.central_act_dim % self.output_dim != 0
        self.num_agents = int(policy_config["num_agents"]) if "num_agents" in policy_config else (
            1 if self.mixed_act_dims else self.central_act_dim // self.output_dim)
        cfg = self.rddpg_cfg(1, 1)
        dev = self.device
        cin = self.central_obs_dim + self.central_act_dim
        mk_a = lambda: R_MADDPG_Actor(a, self.obs_dim, self.act_dim, dev, cfg, values=draw_ractor_values(a, self.obs_dim, self.act_dim))
        mk_c = lambda: R_MADDPG_Critic(a, self.central_obs_dim, self.central_act_dim, dev, cfg, self.num_q,
                                       values=draw_rcritic_values(a, cin, self.num_q))
        # construction order = RNG order of rMADDPGPolicy.py:43-47
        self.actor = mk_a()
        self.critic = mk_c()
        self.target_actor = mk_a()
        self.target_critic = mk_c()
        self.hard_target_updates()
        if train:
            self.actor_optimizer = FlatAdam(self.actor.padded_numel, self.lr, self.opti_eps, dev)
            self.critic_optimizer = FlatAdam(self.critic.padded_numel, self.lr, self.opti_eps, dev)
            self.exploration = DecayThenFlatSchedule(a.epsilon_start, a.epsilon_finish, a.epsilon_anneal_time, decay="linear")

    def rddpg_cfg(self, batch, episode_length):
        a = self.args
        cfg = _lib.RddpgCfg()
        cfg.dims = _lib.Dims(self.num_agents, self.output_dim, self.obs_dim, self.central_obs_dim, int(episode_length))
        if self.mixed_act_dims:
            cfg.joint_act_dim, cfg.joint_act_col = int(self.central_act_dim), 0
        if self.multidiscrete:      # one-hot blocks, argmax / gumbel-softmax per block (ope_rddpg_cfg.n_act_heads)
            cfg.n_act_heads = len(self.act_dim)
            for i, a_dim in enumerate(self.act_dim):
                cfg.act_head_dims[i] = int(a_dim)
        cfg.batch, cfg.num_q = int(batch), self.num_q
        cfg.continuous = int(not self.discrete)
        cfg.target_gumbel = int(self.target_noise is not None and self.discrete)
        cfg.use_huber, cfg.use_per = int(bool(a.use_huber_loss)), int(bool(a.use_per))
        cfg.gamma, cfg.huber_delta = float(a.gamma), float(a.huber_delta)
        return cfg

    # ---- rollout-side API (host logic around the HIP actor forward) ---------------------------------------
    def get_actions(self, obs, prev_actions, rnn_states, available_actions=None, t_env=None, explore=False, use_target=False,
                    use_gumbel=False):
        """rMADDPGPolicy.py:61-131 (discrete branch)."""
        obs = np.asarray(obs) if not torch.is_tensor(obs) else obs
        no_sequence = len(obs.shape) == 2
        batch_size = obs.shape[0] if no_sequence else obs.shape[1]
        actor_out, new_rnn_states = (self.target_actor if use_target else self.actor)(obs, prev_actions, rnn_states)
        actions, eps = self._actions_from_actor_out(actor_out, batch_size, no_sequence, available_actions, t_env, explore, use_target, use_gumbel)
        return actions, new_rnn_states, eps

    def _actions_from_actor_out(self, actor_out, batch_size, no_sequence, available_actions=None, t_env=None, explore=False, use_target=False,
                                use_gumbel=False):
        """Everything of get_actions behind the actor network (rMADDPGPolicy.py:81-131): host logic, the numpy and torch generators consumed
        in the reference's order (pinned by tests/test_rollout_actions.py on the reference's own outputs)."""
        eps = None
        if not self.discrete:      # rMADDPGPolicy.py:121-129
            from ...maddpg.algorithm.MADDPGPolicy import gaussian_noise
            if explore:
                assert no_sequence, "Cannot do exploration on a sequence!"
                actions = gaussian_noise(actor_out.shape, self.args.act_noise_std).to(actor_out.device) + actor_out
            elif use_target and self.target_noise is not None:
                assert isinstance(self.target_noise, float)
                actions = gaussian_noise(actor_out.shape, self.target_noise).to(actor_out.device) + actor_out
            else:
                actions = actor_out
            return actions, eps
        if self.multidiscrete:      # rMADDPGPolicy.py:81-102: every sub-action on its own, no availability masks
            outs = torch.split(actor_out, [int(x) for x in self.act_dim], dim=-1)
            if use_gumbel or (use_target and self.target_noise is not None):
                actions = torch.cat([gumbel_softmax_hard(o, None, sample_gumbel_uniform(o.shape)) for o in outs], dim=-1)
            elif explore:
                assert no_sequence, "Cannot do exploration on a sequence!"
                onehot = torch.cat([gumbel_softmax_hard(o, None, sample_gumbel_uniform(o.shape)) for o in outs], dim=-1)
                eps = self.exploration.eval(t_env)
                rand_numbers = np.random.rand(batch_size, 1)
                take_random = (rand_numbers < eps).astype(int).reshape(-1, 1)
                random_actions = torch.cat([OneHotCategorical(logits=torch.ones(batch_size, int(x))).sample() for x in self.act_dim], dim=1)
                actions = (1 - take_random) * onehot.detach().cpu().numpy() + take_random * random_actions.numpy()
            else:
                actions = torch.cat([onehot_from_logits(o) for o in outs], dim=-1)
            return actions, eps
        if use_gumbel or (use_target and self.target_noise is not None):
            actions = gumbel_softmax_hard(actor_out, available_actions, sample_gumbel_uniform(actor_out.shape))
        elif explore:
            onehot = gumbel_softmax_hard(actor_out, available_actions, sample_gumbel_uniform(actor_out.shape))
            assert no_sequence, "Cannot do exploration on a sequence!"
            eps = self.exploration.eval(t_env)
            rand_numbers = np.random.rand(batch_size, 1)
            logits = torch.ones(batch_size, self.act_dim)
            if available_actions is not None:
                logits[torch.as_tensor(np.asarray(available_actions)) == 0] = -1e10
            random_actions = OneHotCategorical(logits=logits).sample().numpy()
            take_random = (rand_numbers < eps).astype(int)
            actions = (1 - take_random) * onehot.detach().cpu().numpy() + take_random * random_actions
        else:
            actions = onehot_from_logits(actor_out, available_actions)
        return actions, eps

    def init_hidden(self, num_agents, batch_size):
        if num_agents == -1:
            return torch.zeros(batch_size, self.hidden_size)
        return torch.zeros(num_agents, batch_size, self.hidden_size)

    def get_random_actions(self, obs, available_actions=None):
        batch_size = obs.shape[0]
        if not self.discrete:      # rMADDPGPolicy.py:158-159
            return np.random.uniform(self.act_space.low, self.act_space.high, size=(batch_size, self.act_dim))
        if self.multidiscrete:     # rMADDPGPolicy.py:147-151
            return np.concatenate([OneHotCategorical(logits=torch.ones(batch_size, int(x))).sample().numpy() for x in self.act_dim], axis=-1)
        logits = torch.ones(batch_size, self.act_dim)
        if available_actions is not None:
            logits[torch.as_tensor(np.asarray(available_actions)) == 0] = -1e10
        return OneHotCategorical(logits=logits).sample().numpy()

    # ---- target updates (rMADDPGPolicy.py:166-176) -----------------------------------------------------------
    def _polyak(self, tau):
        st = _lib.current_stream()
        _lib.check(_lib.lib.ope_polyak(self.critic.padded_numel, _lib.ptr(self.critic._flat), _lib.ptr(self.target_critic._flat),
                                       float(tau), st), "ope_polyak")
        _lib.check(_lib.lib.ope_polyak(self.actor.padded_numel, _lib.ptr(self.actor._flat), _lib.ptr(self.target_actor._flat),
                                       float(tau), st), "ope_polyak")

    def soft_target_updates(self):
        self._polyak(self.args.tau)

    def hard_target_updates(self):
        self._polyak(1.0)
