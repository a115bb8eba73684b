"""Which fields of the batch cost the gather launch what: the same launch with subsets of the seven fields (null pointers = field not copied),
per-dispatch kernel durations (ope_store_gather_profile), HBM-resident store.   python tools/gather_parts.py [episodes=5000] [B=32]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd import _lib
from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_fill_device
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer

NEP = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dims = DIMS["3s5z"]
buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, NEP, dims.episode_length, True, True, device="cuda:0")
pb = buf.policy_buffers["policy_0"]
synth_fill_device(pb, NEP, dims, seed=100, avail="bernoulli")
out = pb.alloc_batch(B)
ALL = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")
sets = [("all seven", ALL), ("obs", ("obs",)), ("obs + share_obs", ("obs", "share_obs")), ("the five short-row fields", ALL[2:]),
        ("acts + avail_acts", ("acts", "avail_acts")), ("rewards + dones + dones_env", ("rewards", "dones", "dones_env")), ("share_obs", ("share_obs",))]
T, N, D, S, A = dims.episode_length, dims.n_agents, dims.obs_dim, dims.state_dim, dims.act_dim
per_ep = dict(obs=(T + 1) * N * D, share_obs=(T + 1) * S, acts=T * N * A, rewards=T * N, dones=T * N, dones_env=T, avail_acts=(T + 1) * N * A)
rng = np.random.RandomState(0)
for name, keep in sets:
    sf, of = pb._store_fields(), pb._fields(out)
    for k in ALL:
        if k not in keep:
            setattr(sf, k, None); setattr(of, k, None)
    _lib.check(_lib.lib.ope_store_gather_profile(1), "profile")
    for _ in range(40):
        inds = np.ascontiguousarray(rng.choice(NEP, B), dtype=np.int64)
        _lib.check(_lib.lib.ope_store_gather_host_inds(C.byref(pb.dims), pb.buffer_size, C.byref(sf), inds.ctypes.data_as(C.c_void_p), B, C.byref(of),
                                                      _lib.current_stream()), "gather")
    torch.cuda.synchronize()
    bufm = (C.c_float * 512)()
    n = _lib.lib.ope_store_gather_profile_read(bufm, 512)
    ms = np.array([bufm[i] for i in range(n)][8:])
    _lib.check(_lib.lib.ope_store_gather_profile(0), "profile")
    mb = 2.0 * 4 * B * sum(per_ep[k] for k in keep) / 1e6
    print("%-34s %6.2f MB  %6.2f us  (min %5.2f)  %5.2f TB/s" % (name, mb, 1e3 * np.median(ms), 1e3 * ms.min(), mb / (1e3 * np.median(ms))))
