"""MLP VDN policy = MLP QMIX policy (offpolicy/algorithms/mvdn/algorithm/mVDNPolicy.py:3-5)."""
from ...mqmix.algorithm.mQMixPolicy import M_QMixPolicy


class M_VDNPolicy(M_QMixPolicy):
    pass
