"""Micro-benchmark of ope_store_gather alone (HIP events around the launches). Usage: python tools/bench_gather.py [workload] [B]"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd import _lib
from offpolicy_amd.utils.synth import DIMS, policy_info_for
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
wl = sys.argv[1] if len(sys.argv) > 1 else "3s5z"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dims = DIMS[wl]
buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, 256, dims.episode_length, True, True, device="cuda:0")
pb = buf.policy_buffers["policy_0"]
pb._ring.filled_i = 256
for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts"):
    getattr(pb, k).normal_()
rng = np.random.RandomState(0)
n = 200
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for i in range(20):
    pb.sample_inds(rng.randint(0, 256, B))
for i in range(n):
    pb.sample_inds(rng.randint(0, 256, B), timing_events=ev[i])
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in ev])
byt = 2.0 * B * _lib.lib.ope_episode_bytes(C.byref(pb.dims))
print("gather %s B=%d FLOATS=%s TMAJOR=%s: median %.2f us  min %.2f us  -> %.0f GB/s (median), %.0f GB/s (best)" % (
    wl, B, os.environ.get("OPE_GATHER_FLOATS", "def"), os.environ.get("OPE_GATHER_TMAJOR", "0"),
    1e3 * np.median(ms), 1e3 * ms.min(), byt / np.median(ms) / 1e6, byt / ms.min() / 1e6))
