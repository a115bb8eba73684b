"""Per-wave phase timing of mixer_fwd2 from s_memtime stamps (ope_set_debug(1)). Run with OPE_MIXER_PERSIST=0 at 3s5z-sized
states (the default there is mixer_fwd3: tools/mixer3_phases.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd import _lib
from offpolicy_amd.config import default_args
from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_episodes
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
from offpolicy_amd.algorithms.qmix.qmix import QMix
dims = DIMS[os.environ.get("OPE_PHASE_DIMS", "3s5z")]; B = int(os.environ.get("OPE_PHASE_B", "32"))
dev = torch.device("cuda:0"); pinfo = policy_info_for(dims)
policy = QMixPolicy({"args": default_args(), "device": dev}, pinfo["policy_0"])
trainer = QMix(default_args(), dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length)
buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, 64, dims.episode_length, True, True, device=dev)
ep = synth_episodes(np.random.RandomState(0), 64, dims, avail="bernoulli")
buf.insert(64, *[{"policy_0": ep[k]} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")])
s = buf.policy_buffers["policy_0"].sample_inds(np.arange(B))
batch = tuple({"policy_0": x} for x in s) + (None, None)
_lib.lib.ope_set_debug(1)
for _ in range(3):
    trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
trainer.workspace_view(B, "dbg").zero_()          # stale stamps of other kernels / earlier steps would pollute the table
trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
nw = 2 * ((dims.episode_length * B + 15) // 16) * 4   # waves of mixer_fwd2 (RT = 1): 2 nets x row tiles x 4
d = trainer.workspace_view(B, "dbg").view(torch.int64).cpu().numpy()[:8 * nw].reshape(-1, 8)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
rel = (d[:, :6] - t0).astype(np.float64)
print("waves", len(d), "clock ticks; kernel span", rel[:, :5].max())
names = ["start", "afterA", "bar1", "afterB", "bar2", "end(w1)"]
for i, n in enumerate(names):
    col = rel[:, i][d[:, i] > 0]
    print("%-8s min %9.0f  median %9.0f  p90 %9.0f  max %9.0f" % (n, col.min(), np.median(col), np.percentile(col, 90), col.max()))
dur = d[:, 4] - d[:, 0]
print("per-wave lifetime: median %.0f p90 %.0f max %.0f ; stageA median %.0f ; bar1 wait median %.0f ; stageB median %.0f" % (
    np.median(dur), np.percentile(dur, 90), dur.max(), np.median(d[:, 1] - d[:, 0]), np.median(d[:, 2] - d[:, 1]), np.median(d[:, 3] - d[:, 2])))
