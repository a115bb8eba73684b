"""Host-side cost of enqueueing one bench step (sample + train), measured right after a synchronize while the device queue is short:
the time the host needs per step decides whether it stays ahead of the device (0.3 ms of kernels per step at 3s5z).
    python tools/host_step_time.py [--profile]
"""
import sys, os, time, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd.config import default_args
from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_fill_device
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
from offpolicy_amd.algorithms.qmix.qmix import QMix
dims = DIMS["3s5z"]; args = default_args(); dev = torch.device("cuda:0")
pinfo = policy_info_for(dims)
policy = QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
with contextlib.redirect_stdout(sys.stderr):
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda x: "policy_0", device=dev, episode_length=dims.episode_length)
trainer.fuse_soft_update = True
E = 512
buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, E, dims.episode_length, True, True, device=dev)
pbuf = buf.policy_buffers["policy_0"]
synth_fill_device(pbuf, E, dims, seed=100, avail="bernoulli")
def one_step(live):
    inds = np.random.choice(E, 32)
    s = pbuf.sample_inds(inds, live_for=trainer if live else None)
    info, _, _ = trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s) + (None, None))
    trainer.soft_target_updates()
for live in (True, False):
    for _ in range(30): one_step(live)
    torch.cuda.synchronize()
    ts = []
    for rep in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12): one_step(live)
        ts.append((time.perf_counter() - t0) / 12)
    torch.cuda.synchronize()
    print("live_for=%s: host enqueue time per step: median %.1f us (min %.1f)" % (live, 1e6 * np.median(ts), 1e6 * min(ts)))
if "--profile" in sys.argv:
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): one_step(True)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
