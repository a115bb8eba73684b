"""Parameter-free sum mixer (offpolicy/algorithms/vdn/algorithm/vdn_mixer.py:6-40); the sum itself is fused into the TD
kernel (`vdn_kernel`, csrc/ope_mixer.hip)."""
from ...qmix.algorithm.q_mixer import VDNMixer  # noqa: F401
