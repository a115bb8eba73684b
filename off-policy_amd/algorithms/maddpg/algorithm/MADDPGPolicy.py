"""MADDPG / MATD3 policy: actor + centralised critic (+ targets, optimizer state) on flat CUDA vectors.

Mirror of offpolicy/algorithms/maddpg/algorithm/MADDPGPolicy.py:11-151 for discrete (one-hot) and, round 4, continuous (Box) action
spaces (an action = the actor's output, MADDPGPolicy.py:107-116) and multi-discrete ones (MADDPGPolicy.py:73-92: one argmax / hard
gumbel-softmax per sub-action block of the one stacked head; up to six sub-actions). The four
networks are drawn in the reference's construction order (actor, critic, target actor, target critic) so equal seeds
give equal weights -- including the target critic's own random Q heads, which upstream never synchronises (A-4).
"""
import numpy as np
import torch
from torch.distributions import OneHotCategorical

from .... import _lib
from ....config import require_reference_architecture
from ....utils.spaces import get_dim_from_space
from ...qmix.algorithm.QMixPolicy import DecayThenFlatSchedule
from ...qmix.qmix import FlatAdam
from .actor_critic import MADDPG_Actor, MADDPG_Critic, draw_actor_values, draw_critic_values


def sample_gumbel_uniform(shape):
    """The uniform draw inside sample_gumbel (util.py:178-181): CPU generator, torch.FloatTensor(*shape).uniform_()."""
    return torch.FloatTensor(*shape).uniform_()


def gaussian_noise(shape, std):
    """util.py:217-218: CPU generator, torch.empty(shape).normal_(mean=0, std=std)."""
    return torch.empty(shape).normal_(mean=0, std=std)


def gumbel_uniform_for(policy, rows):
    """The uniform noise of one hard gumbel-softmax over `rows` rows of the policy's action vector, drawn as the reference draws it: one
    torch.FloatTensor(rows, a).uniform_() per (sub-)action head in order (util.py:178-190; MADDPGPolicy.py:75-77), concatenated."""
    if getattr(policy, "multidiscrete", False):
        return torch.cat([sample_gumbel_uniform((rows, int(a))) for a in policy.act_dim], dim=-1)
    return sample_gumbel_uniform((rows, policy.output_dim))


def target_noise_for(policy, rows):
    """What policy.get_actions(use_target=True) draws for `rows` rows, on the generator the reference uses (MADDPGPolicy.py:73-116): nothing
    without a target noise (MADDPG), the uniform block(s) of a hard gumbel-softmax for discrete / multi-discrete actions, additive gaussian
    noise for continuous ones."""
    if policy.target_noise is None:
        return None
    if policy.discrete:
        return gumbel_uniform_for(policy, rows)
    return gaussian_noise((rows, policy.output_dim), float(policy.target_noise))


def onehot_from_logits(logits, avail=None):
    logits = logits.clone()
    if avail is not None:
        logits[torch.as_tensor(np.asarray(avail), device=logits.device) == 0] = -1e10
    return (logits == logits.max(-1, keepdim=True)[0]).float()


def gumbel_softmax_hard(logits, avail, u):
    y = logits + (-torch.log(-torch.log(u.to(logits.device) + 1e-20) + 1e-20))
    if avail is not None:
        y[torch.as_tensor(np.asarray(avail), device=y.device) == 0] = -1e10
    y = torch.softmax(y, dim=-1)
    y_hard = (y == y.max(-1, keepdim=True)[0]).float()
    return (y_hard - y) + y


class MADDPGPolicy(object):
    def __init__(self, config, policy_config, target_noise=None, td3=False, train=True, frozen_q_head=True):
        self.config = config
        self.device = torch.device(config["device"])
        self.args = config["args"]
        require_reference_architecture(self.args)
        self.tau, self.lr, self.opti_eps, self.weight_decay = self.args.tau, self.args.lr, self.args.opti_eps, self.args.weight_decay
        self.central_obs_dim, self.central_act_dim = policy_config["cent_obs_dim"], policy_config["cent_act_dim"]
        self.obs_space, self.act_space = policy_config["obs_space"], policy_config["act_space"]
        self.obs_dim, self.act_dim = get_dim_from_space(self.obs_space), get_dim_from_space(self.act_space)
        kind = self.act_space.__class__.__name__
        self.discrete, self.multidiscrete = kind != "Box", "MultiDiscrete" in kind      # util.py:271-281 is_discrete / is_multidiscrete
        # act_dim: as upstream an int, or the ARRAY of a multi-discrete space's sub-action sizes; output_dim: the action vector's length
        self.output_dim = int(sum(self.act_dim)) if self.multidiscrete else self.act_dim
        if self.multidiscrete and len(self.act_dim) > 6:
            raise NotImplementedError("multi-discrete action spaces with more than 6 sub-actions are not on the accelerated path")
        self.target_noise = target_noise
        self.td3 = bool(td3)
        self.num_q = 2 if td3 else 1
        # the joint action is this policy's width times the number of agents, unless policies of OTHER action dimensions share the critic
        # (share_policy = False on e.g. simple_speaker_listener): then only its total width is known here (ope_ddpg_cfg.joint_act_dim)
        # `policy_config["num_agents"]` (optional, what a caller that knows it should pass) settles it; without it the guess below holds for one
        # shared policy and can be WRONG for policies of different widths that happen to divide the total (2 and 4 of 8) -- the trainers
        # overwrite dims.n_agents / joint_act_dim with the real layout before any launch, so nothing on the update path reads the guess
        self.mixed_act_dims = self.central_act_dim % self.output_dim != 0
        self.num_agents = int(policy_config["num_agents"]) if "num_agents" in policy_config else (
            1 if self.mixed_act_dims else self.central_act_dim // self.output_dim)
        self.frozen_q_head = bool(frozen_q_head)
        cfg = self.ddpg_cfg(1)
        dev, a = self.device, self.args
        mk_c = lambda vals: MADDPG_Critic(a, self.central_obs_dim, self.central_act_dim, dev, cfg, self.num_q, values=vals,
                                          frozen_q_head=self.frozen_q_head)
        # construction order = RNG order of MADDPGPolicy.py:38-46
        self.actor = MADDPG_Actor(a, self.obs_dim, self.act_dim, dev, cfg, values=draw_actor_values(a, self.obs_dim, self.act_dim))
        self.critic = mk_c(draw_critic_values(a, self.central_obs_dim + self.central_act_dim, self.num_q))
        self.target_actor = MADDPG_Actor(a, self.obs_dim, self.act_dim, dev, cfg, values=draw_actor_values(a, self.obs_dim, self.act_dim))
        self.target_critic = mk_c(draw_critic_values(a, self.central_obs_dim + self.central_act_dim, self.num_q))
        self.hard_target_updates()          # load_state_dict of the REGISTERED tensors (heads stay apart when frozen)
        if train:
            self.actor_optimizer = FlatAdam(self.actor.padded_numel, self.lr, self.opti_eps, dev)
            self.critic_optimizer = FlatAdam(self.critic.padded_numel, self.lr, self.opti_eps, dev)
            self.exploration = DecayThenFlatSchedule(a.epsilon_start, a.epsilon_finish, a.epsilon_anneal_time, decay="linear")

    def ddpg_cfg(self, batch):
        a = self.args
        cfg = _lib.DdpgCfg()
        cfg.dims = _lib.Dims(self.num_agents, self.output_dim, self.obs_dim, self.central_obs_dim, 1)
        if self.mixed_act_dims:
            cfg.joint_act_dim, cfg.joint_act_col = int(self.central_act_dim), 0
        if self.multidiscrete:      # the action vector = one-hot blocks, argmax / gumbel-softmax per block (ope_ddpg_cfg.n_act_heads)
            cfg.n_act_heads = len(self.act_dim)
            for i, a_dim in enumerate(self.act_dim):
                cfg.act_head_dims[i] = int(a_dim)
        cfg.batch, cfg.num_q = int(batch), self.num_q
        cfg.continuous = int(not self.discrete)
        cfg.target_gumbel = int(self.target_noise is not None and self.discrete)
        cfg.use_huber, cfg.use_per = int(bool(a.use_huber_loss)), int(bool(a.use_per))
        cfg.gamma, cfg.huber_delta, cfg.per_eps = float(a.gamma), float(a.huber_delta), float(a.per_eps)
        return cfg

    # ---- rollout-side API (host logic around the HIP actor forward) ---------------------------------------
    def get_actions(self, obs, available_actions=None, t_env=None, explore=False, use_target=False, use_gumbel=False):
        actor_out = (self.target_actor if use_target else self.actor)(obs)
        return self._actions_from_actor_out(actor_out, obs.shape[0], available_actions, t_env, explore, use_target, use_gumbel)

    def _actions_from_actor_out(self, actor_out, batch_size, available_actions=None, t_env=None, explore=False, use_target=False, use_gumbel=False):
        """Everything of get_actions behind the actor network (MADDPGPolicy.py:72-119): host logic, the numpy and torch generators consumed in
        the reference's order (pinned by tests/test_rollout_actions.py on the reference's own outputs)."""
        eps = None
        if not self.discrete:      # MADDPGPolicy.py:107-116
            if explore:
                actions = gaussian_noise(actor_out.shape, self.args.act_noise_std).to(actor_out.device) + actor_out
            elif use_target and self.target_noise is not None:
                assert isinstance(self.target_noise, float)
                actions = gaussian_noise(actor_out.shape, self.target_noise).to(actor_out.device) + actor_out
            else:
                actions = actor_out
            return actions, eps
        if self.multidiscrete:      # MADDPGPolicy.py:73-92: every sub-action on its own, no availability masks
            outs = torch.split(actor_out, [int(x) for x in self.act_dim], dim=-1)
            if use_gumbel or (use_target and self.target_noise is not None):
                actions = torch.cat([gumbel_softmax_hard(o, None, sample_gumbel_uniform(o.shape)) for o in outs], dim=-1)
            elif explore:
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

    def get_random_actions(self, obs, available_actions=None):
        batch_size = obs.shape[0]
        if not self.discrete:      # MADDPGPolicy.py:135-136
            return np.random.uniform(self.act_space.low, self.act_space.high, size=(batch_size, self.act_dim))
        if self.multidiscrete:     # MADDPGPolicy.py:126-129
            return np.concatenate([OneHotCategorical(logits=torch.ones(batch_size, int(x))).sample().numpy() for x in self.act_dim], axis=-1)
        logits = torch.ones(batch_size, self.act_dim)
        if available_actions is not None:
            logits[torch.as_tensor(np.asarray(available_actions)) == 0] = -1e10
        return OneHotCategorical(logits=logits).sample().numpy()

    # ---- target updates (MADDPGPolicy.py:141-151) ------------------------------------------------------------
    def _polyak(self, tau):
        st = _lib.current_stream()
        _lib.check(_lib.lib.ope_polyak(self.critic.trainable_numel, _lib.ptr(self.critic._flat), _lib.ptr(self.target_critic._flat),
                                       float(tau), st), "ope_polyak")
        _lib.check(_lib.lib.ope_polyak(self.actor.padded_numel, _lib.ptr(self.actor._flat), _lib.ptr(self.target_actor._flat),
                                       float(tau), st), "ope_polyak")

    def soft_target_updates(self):
        if getattr(self, "_polyak_done", False):     # already applied inside the trainer's Adam kernels for this update
            self._polyak_done = False
            return
        self._polyak(self.args.tau)

    def hard_target_updates(self):
        self._polyak(1.0)
