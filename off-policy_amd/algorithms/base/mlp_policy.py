"""Abstract feed-forward policy interface (offpolicy/algorithms/base/mlp_policy.py:4-30)."""
from abc import ABC, abstractmethod


class MLPPolicy(ABC):
    @abstractmethod
    def get_actions(self, obs, available_actions, t_env, explore):
        raise NotImplementedError

    @abstractmethod
    def get_random_actions(self, obs, available_actions):
        raise NotImplementedError
