"""Abstract trainer interface (offpolicy/algorithms/base/trainer.py:4-37): what the runners call on a trainer."""
from abc import ABC, abstractmethod


class Trainer(ABC):
    @abstractmethod
    def __init__(self, args, num_agents, policies, policy_mapping_fn, device, episode_length):
        raise NotImplementedError

    @abstractmethod
    def train_policy_on_batch(self, update_policy_id, batch):
        """One update of `update_policy_id` on a sampled batch -> (train_info, new_priorities, idxes)."""
        raise NotImplementedError

    @abstractmethod
    def prep_training(self):
        raise NotImplementedError

    @abstractmethod
    def prep_rollout(self):
        raise NotImplementedError
