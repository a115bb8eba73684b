"""VDN policy = QMIX policy (offpolicy/algorithms/vdn/algorithm/VDNPolicy.py:3-5)."""
from ...qmix.algorithm.QMixPolicy import QMixPolicy


class VDNPolicy(QMixPolicy):
    pass
