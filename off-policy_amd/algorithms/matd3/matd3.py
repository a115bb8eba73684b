"""MATD3 = MADDPG with twin Q heads, target-action gumbel noise and (intended) delayed actor updates
(offpolicy/algorithms/matd3/matd3.py:3-5: actor_update_interval=2)."""
from ..maddpg.maddpg import MADDPG


class MATD3(MADDPG):
    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=None, count_updates=False):
        super(MATD3, self).__init__(args, num_agents, policies, policy_mapping_fn, device=device, actor_update_interval=2,
                                    count_updates=count_updates)
