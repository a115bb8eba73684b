"""MLP VDN trainer = M_QMix with sum mixing (offpolicy/algorithms/mvdn/mvdn.py:4-6). The reference's M_VDNMixer.forward
takes one argument but is called with two (TypeError); here, as in the patched oracle, the state
argument is accepted and ignored."""
from ..mqmix.mqmix import M_QMix


class M_VDN(M_QMix):
    def __init__(self, args, num_agents, policies, policy_mapping_fn, device=None):
        super(M_VDN, self).__init__(args, num_agents, policies, policy_mapping_fn, device, vdn=True)
