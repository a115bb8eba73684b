"""Parameter-free sum mixer of the MLP family (offpolicy/algorithms/mvdn/algorithm/mvdn_mixer.py); fused into the TD kernel."""
from ...qmix.algorithm.q_mixer import VDNMixer as M_VDNMixer  # noqa: F401
