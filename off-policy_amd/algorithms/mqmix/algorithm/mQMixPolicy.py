"""MLP QMIX / VDN policy (mirror of offpolicy/algorithms/mqmix/algorithm/mQMixPolicy.py:8-136)."""
import numpy as np
import torch
from torch.distributions import Categorical, OneHotCategorical

from ....config import require_reference_architecture
from ....utils.spaces import get_dim_from_space
from ...qmix.algorithm.QMixPolicy import DecayThenFlatSchedule
from .agent_q_function import AgentQFunction


class M_QMixPolicy(object):
    def __init__(self, config, policy_config, train=True):
        self.args = config["args"]
        self.device = torch.device(config["device"])
        require_reference_architecture(self.args)
        self.obs_space = policy_config["obs_space"]
        self.obs_dim = get_dim_from_space(self.obs_space)
        self.act_space = policy_config["act_space"]
        self.act_dim = get_dim_from_space(self.act_space)
        if np.ndim(self.act_dim) != 0 or self.act_space.__class__.__name__ == "Box":
            # upstream gives a MultiDiscrete space one Q head per sub-action and the mixer one input per (agent, sub-action)
            # (QMixPolicy.py:76-93, qmix.py:49-57); the kernels carry one head per agent. Box spaces are not Q-learning's (upstream asserts)
            raise NotImplementedError("the Q-learning families take Discrete action spaces on the accelerated path (got %s)" % self.act_space.__class__.__name__)
        self.output_dim = self.act_dim
        self.hidden_size = self.args.hidden_size
        self.central_obs_dim = policy_config["cent_obs_dim"]
        self.discrete_action, self.multidiscrete = True, False
        self.q_network_input_dim = self.obs_dim
        self.q_network = AgentQFunction(self.args, self.q_network_input_dim, self.act_dim, self.device)
        if train:
            self.exploration = DecayThenFlatSchedule(self.args.epsilon_start, self.args.epsilon_finish,
                                                     self.args.epsilon_anneal_time, decay="linear")

    def get_q_values(self, obs_batch, action_batch=None):
        q_batch = self.q_network(obs_batch)
        if action_batch is not None:
            action_batch = torch.as_tensor(action_batch).to(q_batch.device).long()
            return torch.gather(q_batch, 1, action_batch.unsqueeze(dim=-1))
        return q_batch

    def get_actions(self, obs_batch, available_actions=None, t_env=None, explore=False):
        batch_size = obs_batch.shape[0]
        q_values = self.get_q_values(obs_batch)
        if available_actions is not None:
            q_values = q_values.clone()
            q_values[torch.as_tensor(np.asarray(available_actions), device=q_values.device) == 0] = -1e10
        greedy_Qs, greedy_actions = q_values.max(dim=-1)
        if explore:
            eps = self.exploration.eval(t_env)
            rand_numbers = np.random.rand(batch_size)
            logits = torch.ones(batch_size, self.act_dim)
            if available_actions is not None:
                logits[torch.as_tensor(np.asarray(available_actions)) == 0] = -1e10
            random_actions = Categorical(logits=logits).sample().numpy()
            take_random = (rand_numbers < eps).astype(int)
            actions = (1 - take_random) * greedy_actions.detach().cpu().numpy() + take_random * random_actions
            onehot_actions = np.eye(self.act_dim)[actions]
        else:
            greedy_Qs = greedy_Qs.unsqueeze(-1)
            onehot_actions = np.eye(self.act_dim)[greedy_actions.detach().cpu().numpy()]
        return onehot_actions, greedy_Qs

    def get_random_actions(self, obs, available_actions=None):
        batch_size = obs.shape[0]
        logits = torch.ones(batch_size, self.act_dim)
        if available_actions is not None:
            logits[torch.as_tensor(np.asarray(available_actions)) == 0] = -1e10
        return OneHotCategorical(logits=logits).sample().numpy()

    def parameters(self):
        return self.q_network.parameters()

    def load_state(self, source_policy):
        self.q_network.load_state_dict(source_policy.q_network.state_dict())
