"""Micro-benchmark / A-B sweep of ope_store_gather alone (HIP events around the launches, variants interleaved in ONE
process, medians reported) next to a plain contiguous device copy of the same number of bytes (the practical ceiling).

    python tools/bench_gather.py [workload=3s5z] [B=32] [episodes=256] [--sweep]

--sweep runs the knob grid of ope_set_gather_params; without it only the default configuration and the copy ceiling."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd import _lib
from offpolicy_amd.utils.synth import DIMS, policy_info_for
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args[0] if len(args) > 0 else "3s5z"
B = int(args[1]) if len(args) > 1 else 32
NEP = int(args[2]) if len(args) > 2 else 256
sweep = "--sweep" in sys.argv
dims = DIMS[wl]
buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, NEP, dims.episode_length, True, True, device="cuda:0")
pb = buf.policy_buffers["policy_0"]
pb._ring.filled_i = NEP
for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts"):
    getattr(pb, k).normal_()
ep_bytes = int(_lib.lib.ope_episode_bytes(C.byref(pb.dims)))
byt = 2.0 * B * ep_bytes
DEFAULT = dict(floats=2048, xcd=8, unroll=8, nt=0, small=1, tile=2048)
cfgs = [("default", {}), ("default, device-resident indices", dict(dev=1))]
if sweep:
    cfgs += [("small=0 (short rows through the step path)", dict(small=0)), ("tile 1024", dict(tile=1024)), ("tile 4096", dict(tile=4096)),
             ("tile 7168", dict(tile=7168)), ("nt loads", dict(nt=1)), ("nt stores", dict(nt=2)), ("nt both", dict(nt=3)),
             ("unroll 4", dict(unroll=4)), ("unroll 16", dict(unroll=16)), ("floats 1024", dict(floats=1024)),
             ("floats 4096", dict(floats=4096)), ("floats 8192", dict(floats=8192)), ("xcd off", dict(xcd=1)),
             ("small=0 + floats 4096", dict(small=0, floats=4096))]


def apply(over):
    c = dict(DEFAULT)
    c.update({k: v for k, v in over.items() if k != "dev"})
    # per store, with the call (ope_gather_tune): no process state
    pb.gather_tune = dict(floats_per_block=c["floats"], xcd_run=c["xcd"], unroll=c["unroll"], nontemporal=1 + c["nt"], small_tiles=1 if c["small"] else 2,
                          tile_floats=c["tile"])


rng = np.random.RandomState(0)
out = pb.alloc_batch(B)
ref = None
for name, over in cfgs:            # every variant returns the same bytes
    apply(over)
    got = pb.sample_inds(np.arange(B) * (NEP // B), out=pb.alloc_batch(B))
    torch.cuda.synchronize()
    if ref is None:
        ref = [g.clone() if g is not None else None for g in got]
    else:
        for a, b in zip(ref, got):
            assert (a is None and b is None) or torch.equal(a, b), name

n_copy = int(byt // 2 // 4)
src = torch.randn(n_copy, device="cuda:0")
dst = torch.empty_like(src)
ROUNDS, PER = 6, 25
times = {name: [] for name, _ in cfgs}
times["plain contiguous copy (torch copy_, same bytes)"] = []
for r in range(ROUNDS):
    for name, over in cfgs:
        apply(over)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(PER)]
        for i in range(PER):
            ii = rng.randint(0, NEP, B)
            pb.sample_inds(torch.as_tensor(ii, device="cuda:0") if over.get("dev") else ii, timing_events=ev[i], out=out)
        torch.cuda.synchronize()
        times[name] += [a.elapsed_time(b) for a, b in ev[3:]]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(PER)]
    for i in range(PER):
        ev[i][0].record()
        dst.copy_(src)
        ev[i][1].record()
    torch.cuda.synchronize()
    times["plain contiguous copy (torch copy_, same bytes)"] += [a.elapsed_time(b) for a, b in ev[3:]]
apply({})
res = {}
print("gather %s B=%d store=%d episodes (%.2f GB) algorithmic %.1f MB/launch" % (wl, B, NEP, NEP * ep_bytes / 1e9, byt / 1e6))
for name, t in times.items():
    t = np.asarray(t)
    med, mn = float(np.median(t)), float(t.min())
    res[name] = dict(median_us=round(1e3 * med, 2), min_us=round(1e3 * mn, 2), gbs_median=round(byt / med / 1e6, 1), gbs_best=round(byt / mn / 1e6, 1))
    print("  %-48s median %7.2f us  min %7.2f us  -> %6.0f GB/s (median) %6.0f GB/s (best)" % (name, 1e3 * med, 1e3 * mn, byt / med / 1e6, byt / mn / 1e6))
print("JSON " + json.dumps(dict(workload=wl, batch=B, episodes=NEP, algorithmic_bytes=byt, results=res)))
