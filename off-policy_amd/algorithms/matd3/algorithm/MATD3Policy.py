"""MATD3 policy (offpolicy/algorithms/matd3/algorithm/MATD3Policy.py:3-5): twin critics, target noise."""
from ...maddpg.algorithm.MADDPGPolicy import MADDPGPolicy


class MATD3Policy(MADDPGPolicy):
    def __init__(self, config, policy_config, train=True, frozen_q_head=True):
        super(MATD3Policy, self).__init__(config, policy_config, target_noise=config["args"].target_action_noise_std, td3=True,
                                          train=train, frozen_q_head=frozen_q_head)
