"""Per-wave phase timing of trunk_fwd3 (live net) from s_memtime stamps (ope_set_debug(1))."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mixer_phases.py")).read().split("_lib.lib.ope_set_debug(1)")[0])
_lib.lib.ope_set_debug(1)
for _ in range(3):
    trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
d = trainer.workspace_view(B, "dbg").view(torch.int64).cpu().numpy()[16 * 2400:16 * 2400 + 16 * 2048].reshape(-1, 16)
d = d[d[:, 0] > 0]
names = ["start", "w-issued", "p0 loads+sum", "p0 done", "bar", "fc1", "LN1+bar", "fc2+LN2", "wih+st", "t2:p0 loads", "t2:p0 done", "t2:bar", "t2:fc1", "t2:LN1", "t2:fc2LN2", "t2:wih"]
print("waves", len(d))
prev = d[:, 0]
for i in range(1, 16):
    ok = d[:, i] > 0
    dt = (d[:, i] - d[:, i - 1])[ok]
    print("%-14s n=%5d  median %8.0f  p90 %8.0f  max %8.0f   cum median %9.0f" % (names[i], ok.sum(), np.median(dt), np.percentile(dt, 90), dt.max(), np.median((d[:, i] - d[:, 0])[ok])))
