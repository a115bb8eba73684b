"""Per-phase cycle sums of the gru4 scan kernels' compute waves (ope_set_debug(1)): mean shader cycles per step."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mixer_phases.py")).read().split("_lib.lib.ope_set_debug(1)")[0])
_lib.lib.ope_set_debug(1)
for _ in range(3):
    trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
d = trainer.workspace_view(B, "dbg").view(torch.int64).cpu().numpy()
Wf = int(os.environ.get("OPE_GRU4_W", "4")); Wb = int(os.environ.get("OPE_GRU4_W", "4"))   # auto rule at 3s5z (at most 512 rows)
for name, off, n, phases in (("gru_fwd4", 71168, 2 * dims.n_agents * B * Wf, ["reads+FMA", "reduce+gates", "publish", "barrier"]),
                              ("gru_bwd4", 87552, dims.n_agents * B * Wb, ["adjoints", "publish", "barrier", "reads+FMA+reduce"])):
    x = d[off:off + 8 * n].reshape(n, 8).astype(np.float64)
    steps = x[:, 4]
    ok = steps > 0
    per = x[ok, :4] / steps[ok, None]
    print("%s: %d waves, %d steps; cycles per step by phase (mean over waves | min | max):" % (name, ok.sum(), int(steps[ok][0])))
    for i, ph in enumerate(phases):
        print("   %-18s %8.1f | %8.1f | %8.1f" % (ph, per[:, i].mean(), per[:, i].min(), per[:, i].max()))
    print("   %-18s %8.1f   (wave 0 of each row: %s)" % ("total", per.sum(1).mean(), np.round(per.mean(0), 1)))
