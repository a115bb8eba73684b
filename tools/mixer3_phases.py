"""Per-wave phase timing of mixer_fwd3 (resident-weight forward mixer) from s_memtime stamps (ope_set_debug(1))."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mixer_phases.py")).read().split("_lib.lib.ope_set_debug(1)")[0])
_lib.lib.ope_set_debug(1)
for _ in range(3):
    trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
trainer.workspace_view(B, "dbg").zero_()
trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
d = trainer.workspace_view(B, "dbg").view(torch.int64).cpu().numpy()[:256 * 8 * 8].reshape(-1, 8, 8)
d = d[d[:, 0, 0] > 0]
print("workgroups", len(d))
names = ["prologue (weights + first state tile)", "tile 1 stage A", "barrier 1", "tile 1 stage B", "barrier 2", "rest of the wave (tiles 2.. + combine)"]
for w in range(8):
    x = d[:, w, :]
    print("wave %d: " % w + "  ".join("%s %.0f" % (n.split(" (")[0], np.median(x[:, k + 1] - x[:, k])) for k, n in enumerate(names)) +
          "   total %.0f" % np.median(x[:, 6] - x[:, 0]))
