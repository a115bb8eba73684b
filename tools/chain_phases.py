"""Per-wave phase timing of qchain_kernel (ope_chain.hip) from s_memtime stamps (ope_qmix_cfg.debug / ope_set_debug(1))."""
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
print(_lib.last_launches())
ntile = (dims.episode_length * B + 15) // 16
d = trainer.workspace_view(B, "dbg").view(torch.int64).cpu().numpy()[:ntile * 8 * 10].reshape(ntile, 8, 10)
ok = d[:, 0, 0] > 0
d = d[ok]
t0 = d[:, :, 0].min()
names = ["requests issued", "staged + barrier", "heads", "barrier", "mixer stage 2", "combine (2 barriers)", "TD + adjoints", "barrier", "tail (reduce, head adjoint, stores)"]
print("tiles", len(d), " kernel span (ticks) %.0f" % (d[:, :, 9].max() - t0))
start = d[:, 0, 0] - t0
print("workgroup start: min %.0f median %.0f p90 %.0f max %.0f ; workgroup end: median %.0f max %.0f" % (
    start.min(), np.median(start), np.percentile(start, 90), start.max(), np.median(d[:, :, 9].max(1) - t0), (d[:, :, 9].max(1) - t0).max()))
for w in range(8):
    x = d[:, w, :]
    print("wave %d: " % w + "  ".join("%s %.0f" % (n, np.median(x[:, k + 1] - x[:, k])) for k, n in enumerate(names)) + "   total %.0f" % np.median(x[:, 9] - x[:, 0]))
