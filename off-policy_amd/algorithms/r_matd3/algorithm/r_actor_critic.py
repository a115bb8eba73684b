"""R_MATD3 actor / critic (offpolicy/algorithms/r_matd3/algorithm/r_actor_critic.py:3-11): twin Q heads."""
from ...r_maddpg.algorithm.r_actor_critic import R_MADDPG_Actor, R_MADDPG_Critic


class R_MATD3_Actor(R_MADDPG_Actor):
    pass


class R_MATD3_Critic(R_MADDPG_Critic):
    def __init__(self, args, central_obs_dim, central_act_dim, device, cfg, flat=None, values=None):
        super(R_MATD3_Critic, self).__init__(args, central_obs_dim, central_act_dim, device, cfg, num_q_outs=2, flat=flat, values=values)
