"""MI355X-native off-policy MARL update engine (replay sample + QMIX/VDN/MADDPG/MATD3 training step).

Host side mirrors the reference's Python surface (offpolicy.utils.rec_buffer, offpolicy.algorithms.*);
all arithmetic on the path runs in hand-written HIP kernels for gfx950 behind the C-ABI in include/ope.h.
"""
__version__ = "0.1.0"
