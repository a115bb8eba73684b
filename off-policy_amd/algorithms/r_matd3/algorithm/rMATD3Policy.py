"""R_MATD3 policy (offpolicy/algorithms/r_matd3/algorithm/rMATD3Policy.py:3-6): twin critics, target noise."""
from ...r_maddpg.algorithm.rMADDPGPolicy import R_MADDPGPolicy


class R_MATD3Policy(R_MADDPGPolicy):
    def __init__(self, config, policy_config, train=True):
        super(R_MATD3Policy, self).__init__(config, policy_config, target_noise=config["args"].target_action_noise_std, td3=True,
                                            train=train)
