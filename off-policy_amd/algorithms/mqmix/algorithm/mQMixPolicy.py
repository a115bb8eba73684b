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
        require_reference_architecture(self.args, allow_no_feature_norm=True, allow_tanh=True)      # (round 5: the two flags that only touch the first layer / the activation)
        self.obs_space = policy_config["obs_space"]
        self.obs_dim = get_dim_from_space(self.obs_space)
        self.act_space = policy_config["act_space"]
        self.act_dim = get_dim_from_space(self.act_space)
        if self.act_space.__class__.__name__ == "Box":
            raise NotImplementedError("the Q-learning families take Discrete / MultiDiscrete action spaces (upstream asserts the same)")
        # MultiDiscrete (mQMixPolicy.py:21-27, 44-52): one q head per sub-action; see QMixPolicy for how the kernels run it
        self.multidiscrete = np.ndim(self.act_dim) != 0
        self.head_dims = [int(d) for d in np.asarray(self.act_dim).reshape(-1)] if self.multidiscrete else None
        self.output_dim = int(sum(self.head_dims)) if self.multidiscrete else self.act_dim
        self.hidden_size = self.args.hidden_size
        self.central_obs_dim = policy_config["cent_obs_dim"]
        self.discrete_action = not self.multidiscrete
        self.q_network_input_dim = self.obs_dim
        self.q_network = AgentQFunction(self.args, self.q_network_input_dim, self.act_dim, self.device)
        if train:
            self.exploration = DecayThenFlatSchedule(self.args.epsilon_start, self.args.epsilon_finish,
                                                     self.args.epsilon_anneal_time, decay="linear")

    def get_q_values(self, obs_batch, action_batch=None):
        q_batch = self.q_network(obs_batch)
        if self.multidiscrete:      # upstream's network returns the list of per-head q tensors; action_batch = one index vector per head
            q_batch = list(torch.split(q_batch, self.head_dims, dim=-1))
            if action_batch is not None:
                return torch.cat([torch.gather(qb, 1, torch.as_tensor(ab).to(qb.device).long().unsqueeze(dim=-1))
                                  for qb, ab in zip(q_batch, action_batch)], dim=-1)
            return q_batch
        if action_batch is not None:
            action_batch = torch.as_tensor(action_batch).to(q_batch.device).long()
            return torch.gather(q_batch, 1, action_batch.unsqueeze(dim=-1))
        return q_batch

    def get_actions(self, obs_batch, available_actions=None, t_env=None, explore=False):
        return self.actions_from_q(self.get_q_values(obs_batch), obs_batch.shape[0], available_actions, t_env, explore)

    def actions_from_q(self, q_values, batch_size, available_actions=None, t_env=None, explore=False):
        """Everything of get_actions behind the q network (mQMixPolicy.py:62-113): host logic, the reference's generator draw order."""
        if self.multidiscrete:      # mQMixPolicy.py:74-95: per head, in order (one rand + one Categorical draw per head when exploring)
            assert available_actions is None, "MultiDiscrete spaces come without availability masks"
            onehots, greedy = [], []
            for qh, d in zip(q_values, self.head_dims):
                greedy_Q, greedy_action = qh.max(dim=-1)
                if explore:
                    eps = self.exploration.eval(t_env)
                    rand_number = np.random.rand(batch_size)
                    random_action = Categorical(logits=torch.ones(batch_size, d)).sample().numpy()
                    take_random = (rand_number < eps).astype(int)
                    action = (1 - take_random) * greedy_action.detach().cpu().numpy() + take_random * random_action
                    onehots.append(np.eye(d)[action])
                else:
                    greedy_Q = greedy_Q.unsqueeze(-1)
                    onehots.append(np.eye(d)[greedy_action.detach().cpu().numpy()])
                greedy.append(greedy_Q)
            return np.concatenate(onehots, axis=-1), torch.cat(greedy, dim=-1)
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
        if self.multidiscrete:
            return np.concatenate([OneHotCategorical(logits=torch.ones(batch_size, d)).sample().numpy() for d in self.head_dims], axis=-1)
        logits = torch.ones(batch_size, self.act_dim)
        if available_actions is not None:
            logits[torch.as_tensor(np.asarray(available_actions)) == 0] = -1e10
        return OneHotCategorical(logits=logits).sample().numpy()

    def parameters(self):
        return self.q_network.parameters()

    def load_state(self, source_policy):
        self.q_network.load_state_dict(source_policy.q_network.state_dict())
