"""R_MATD3 = R_MADDPG with twin Q heads, target-action gumbel noise and delayed actor updates
(offpolicy/algorithms/r_matd3/r_matd3.py:4-8: actor_update_interval=2; the recurrent trainer DOES count updates)."""
from ..r_maddpg.r_maddpg import R_MADDPG


class R_MATD3(R_MADDPG):
    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=None, episode_length=None):
        super(R_MATD3, self).__init__(args, num_agents, policies, policy_mapping_fn, device=device, episode_length=episode_length,
                                      actor_update_interval=2)
