"""QMIX / VDN policy: per-agent recurrent Q network + epsilon-greedy action selection.

Mirror of offpolicy/algorithms/qmix/algorithm/QMixPolicy.py:9-206. Q values come from the HIP forward
(`AgentQFunction.forward` -> `ope_agent_forward`); the epsilon-greedy bookkeeping around them is host logic in the
same order of RNG draws as the reference (QMixPolicy.py:156-166).
"""
import numpy as np
import torch
from torch.distributions import Categorical, OneHotCategorical

from ....utils.spaces import get_dim_from_space
from ....config import require_reference_architecture
from .agent_q_function import AgentQFunction


class DecayThenFlatSchedule(object):
    """Linear / exponential epsilon schedule (offpolicy/utils/util.py:78-100)."""

    def __init__(self, start, finish, time_length, decay="exp"):
        self.start, self.finish, self.time_length, self.decay = start, finish, time_length, decay
        self.delta = (self.start - self.finish) / self.time_length
        if self.decay == "exp":
            self.exp_scaling = (-1) * self.time_length / np.log(self.finish) if self.finish > 0 else 1

    def eval(self, T):
        if self.decay == "linear":
            return max(self.finish, self.start - self.delta * T)
        return min(self.start, max(self.finish, np.exp(-T / self.exp_scaling)))


def _onehot(idx, dim):
    idx = np.asarray(idx.detach().cpu() if torch.is_tensor(idx) else idx)
    return np.eye(dim)[idx]


class QMixPolicy(object):
    def __init__(self, config, policy_config, train=True):
        self.args = config["args"]
        self.device = torch.device(config["device"])
        require_reference_architecture(self.args, allow_prev_act_inp=True, allow_hypernet_layers_1=True, allow_layer_N_2=True, allow_no_feature_norm=True, allow_tanh=True)   # (the mixer is the trainer's business: QMix checks it)
        self.obs_space = policy_config["obs_space"]
        self.obs_dim = get_dim_from_space(self.obs_space)
        self.act_space = policy_config["act_space"]
        self.act_dim = get_dim_from_space(self.act_space)
        if self.act_space.__class__.__name__ == "Box":
            raise NotImplementedError("the Q-learning families take Discrete / MultiDiscrete action spaces (upstream asserts the same)")
        # MultiDiscrete (QMixPolicy.py:21-27, 76-93): act_dim is the ARRAY of the sub-actions' sizes, the network has one q head per
        # sub-action and the mixer gets one input per (agent, sub-action). The kernels carry one stacked head of sum(act_dim) rows and one q
        # value per "agent": the trainer presents every (agent, sub-action) pair to them as an agent of its own whose availability mask is
        # the sub-action's block (QMix._md_expand); this class does the per-head bookkeeping of the rollout side.
        self.multidiscrete = np.ndim(self.act_dim) != 0
        self.head_dims = [int(d) for d in np.asarray(self.act_dim).reshape(-1)] if self.multidiscrete else None
        self.output_dim = int(sum(self.head_dims)) if self.multidiscrete else self.act_dim
        self.hidden_size = self.args.hidden_size
        self.central_obs_dim = policy_config["cent_obs_dim"]
        self.discrete = not self.multidiscrete
        # previous action as an extra input of the (decentralised) agent network: QMixPolicy.py:29-33
        self.prev_act_inp = bool(getattr(self.args, "prev_act_inp", False))
        if self.multidiscrete and self.prev_act_inp:
            raise NotImplementedError("prev_act_inp with a MultiDiscrete action space (upstream adds the array act_dim to obs_dim there and fails)")
        self.q_network_input_dim = self.obs_dim + self.output_dim if self.prev_act_inp else self.obs_dim
        self.q_network = AgentQFunction(self.args, self.q_network_input_dim, self.act_dim, self.device)
        if train:
            self.exploration = DecayThenFlatSchedule(self.args.epsilon_start, self.args.epsilon_finish,
                                                     self.args.epsilon_anneal_time, decay="linear")

    # -- q values --------------------------------------------------------------------------------------------
    def get_q_values(self, obs_batch, prev_action_batch, rnn_states, action_batch=None):
        if self.prev_act_inp:      # QMixPolicy.py:54-58
            obs_batch = torch.cat((torch.as_tensor(obs_batch, dtype=torch.float32, device=self.device),
                                   torch.as_tensor(prev_action_batch, dtype=torch.float32, device=self.device)), dim=-1)
        q_batch, new_rnn_states = self.q_network(obs_batch, rnn_states)
        if self.multidiscrete:      # upstream's network returns the list of per-head q tensors (act.py:23-31)
            q_batch = list(torch.split(q_batch, self.head_dims, dim=-1))
        if action_batch is not None:
            action_batch = torch.as_tensor(action_batch).to(self.device)
            return self.q_values_from_actions(q_batch, action_batch), new_rnn_states
        return q_batch, new_rnn_states

    def q_values_from_actions(self, q_batch, action_batch):
        if self.multidiscrete:      # QMixPolicy.py:76-87: per head, the q value of that sub-action's one-hot block; side by side
            action_batch = torch.as_tensor(action_batch, device=q_batch[0].device)
            out, ind = [], 0
            for qb, d in zip(q_batch, self.head_dims):
                idx = action_batch[..., ind:ind + d].max(dim=-1)[1]
                out.append(torch.gather(qb, qb.dim() - 1, idx.unsqueeze(dim=-1)))
                ind += d
            return torch.cat(out, dim=-1)
        idx = torch.as_tensor(action_batch, device=q_batch.device).max(dim=-1)[1]
        return torch.gather(q_batch, q_batch.dim() - 1, idx.unsqueeze(dim=-1))

    # -- actions ---------------------------------------------------------------------------------------------
    def get_actions(self, obs, prev_actions, rnn_states, available_actions=None, t_env=None, explore=False):
        q_values_out, new_rnn_states = self.get_q_values(obs, prev_actions, rnn_states)
        onehot_actions, greedy_Qs = self.actions_from_q(q_values_out, available_actions=available_actions,
                                                        explore=explore, t_env=t_env)
        return onehot_actions, new_rnn_states, greedy_Qs

    def actions_from_q(self, q_values, available_actions=None, explore=False, t_env=None):
        if self.multidiscrete:
            return self._actions_from_q_md(q_values, available_actions, explore, t_env)
        no_sequence = q_values.dim() == 2
        batch_size = q_values.shape[0] if no_sequence else q_values.shape[1]
        if available_actions is not None:
            q_values = q_values.clone()
            avail = torch.as_tensor(np.asarray(available_actions) if not torch.is_tensor(available_actions) else available_actions,
                                    device=q_values.device)
            q_values[avail == 0] = -1e10
        greedy_Qs, greedy_actions = q_values.max(dim=-1)
        if explore:
            assert no_sequence, "Can only explore on non-sequences"
            eps = self.exploration.eval(t_env)
            rand_numbers = np.random.rand(batch_size)
            logits = torch.ones(batch_size, self.act_dim)
            if available_actions is not None:
                logits[torch.as_tensor(np.asarray(available_actions.cpu() if torch.is_tensor(available_actions) else available_actions)) == 0] = -1e10
            random_actions = Categorical(logits=logits).sample().numpy()
            take_random = (rand_numbers < eps).astype(int)
            actions = (1 - take_random) * greedy_actions.detach().cpu().numpy() + take_random * random_actions
            onehot_actions = np.eye(self.act_dim)[actions]
        else:
            greedy_Qs = greedy_Qs.unsqueeze(-1)
            onehot_actions = _onehot(greedy_actions, self.act_dim)
        return onehot_actions, greedy_Qs

    def _actions_from_q_md(self, q_values, available_actions, explore, t_env):
        """QMixPolicy.py:112-149: per sub-action head, in order -- greedy index, then (exploring) one np.random.rand(batch) and one
        Categorical(ones).sample() per head, the reference's draw order."""
        assert available_actions is None, "MultiDiscrete spaces come without availability masks (upstream's avail_choose fails on the list of heads)"
        no_sequence = q_values[0].dim() == 2
        batch_size = q_values[0].shape[0] if no_sequence else q_values[0].shape[1]
        onehots, greedy = [], []
        for qh, d in zip(q_values, self.head_dims):
            greedy_Q, greedy_action = qh.max(dim=-1)
            if explore:
                assert no_sequence, "Can only explore on non-sequences"
                eps = self.exploration.eval(t_env)
                rand_number = np.random.rand(batch_size)
                random_action = Categorical(logits=torch.ones(batch_size, d)).sample().numpy()
                take_random = (rand_number < eps).astype(int)
                action = (1 - take_random) * greedy_action.detach().cpu().numpy() + take_random * random_action
                onehots.append(np.eye(d)[action])
            else:
                greedy_Q = greedy_Q.unsqueeze(-1)
                onehots.append(_onehot(greedy_action, d))
            greedy.append(greedy_Q)
        return np.concatenate(onehots, axis=-1), torch.cat(greedy, dim=-1)

    def get_random_actions(self, obs, available_actions=None):
        batch_size = obs.shape[0]
        if self.multidiscrete:      # QMixPolicy.py:181-184: one OneHotCategorical draw per head, in order
            return np.concatenate([OneHotCategorical(logits=torch.ones(batch_size, d)).sample().numpy() for d in self.head_dims], axis=-1)
        logits = torch.ones(batch_size, self.act_dim)
        if available_actions is not None:
            logits[torch.as_tensor(np.asarray(available_actions)) == 0] = -1e10
        return OneHotCategorical(logits=logits).sample().numpy()

    def init_hidden(self, num_agents, batch_size):
        if num_agents == -1:
            return torch.zeros(batch_size, self.hidden_size)
        return torch.zeros(num_agents, batch_size, self.hidden_size)

    def parameters(self):
        return self.q_network.parameters()

    def load_state(self, source_policy):
        self.q_network.load_state_dict(source_policy.q_network.state_dict())
