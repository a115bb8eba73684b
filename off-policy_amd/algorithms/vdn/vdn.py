"""VDN trainer = QMix with sum mixing (offpolicy/algorithms/vdn/vdn.py:4-7)."""
from ..qmix.qmix import QMix


class VDN(QMix):
    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=None, episode_length=None):
        import torch
        super(VDN, self).__init__(args, num_agents, policies, policy_mapping_fn,
                                  device if device is not None else torch.device("cuda:0"), episode_length, vdn=True)
