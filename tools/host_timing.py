"""Where does the HOST spend a QMIX step (eager launches)? perf_counter stamps around the Python calls of one step, no
device synchronisation inside the loop; compares the host's enqueue time per step with the GPU's time per step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mixer_phases.py")).read().split("s = buf.policy_buffers")[0])
pbuf = buf.policy_buffers["policy_0"]
trainer.fuse_soft_update = True
def one(stamps):
    t0 = time.perf_counter()
    inds = np.random.choice(len(buf), B)
    t1 = time.perf_counter()
    s = pbuf.sample_inds(inds)
    t2 = time.perf_counter()
    batch = tuple({"policy_0": x} for x in s) + (None, None)
    info, _, _ = trainer.train_policy_on_batch(batch)
    t3 = time.perf_counter()
    trainer.soft_target_updates()
    t4 = time.perf_counter()
    stamps.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(30):
    one([])
torch.cuda.synchronize()
for n in (20, 200):
    st = []
    t0 = time.perf_counter()
    for _ in range(n):
        one(st)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    a = np.array(st) * 1e6
    print("n=%d: host enqueue %.1f us/step, wall incl. drain %.1f us/step; per call (median us): choice %.1f  sample_inds %.1f  train %.1f  soft %.1f" % (
        n, 1e6 * t_host / n, 1e6 * t_all / n, *np.median(a, axis=0)))
    print("   first 12 steps' train-call times (us):", np.round(a[:12, 2], 0))
