"""Duration of one one-shot all-reduce launch at world = 1 (the vector goes out to the own slot and comes back: push, fence, flag,
poll, fixed-order sum -- everything but the xGMI hop) for the flat-gradient sizes of the bench workloads. HIP events around batches of
launches on the training stream; prints us per launch. `python tools/allreduce_roundtrip.py`"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd.dist import OneShotAllreduce
dev = torch.device("cuda:0")
ar = OneShotAllreduce(dev, 0, 1)
for name, n in (("QMIX 3s5z (S=216)", 118795), ("QMIX 3s5z_gall (S=2232)", 634891), ("QMIX MMM2", 154000), ("MADDPG simple_spread critic", 9000), ("max slot", 1 << 20)):
    x = torch.randn(n, device=dev)
    for _ in range(20):
        ar(x)
    torch.cuda.synchronize()
    best = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ar(x)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 200 * 1e3)
    print("%-30s %8d floats (%6.1f KB): %6.2f us per launch back to back (min of 5 x 200: %.2f, max %.2f)" % (name, n, n * 4 / 1024, sorted(best)[2], min(best), max(best)))
assert not ar.timed_out()
ar.close()
