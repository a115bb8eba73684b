"""Phase timing of mixer_wide_gemm_kernel (ope_mixer_wide.hip) from its per-workgroup stamps (ope_qmix_cfg.debug): cycles spent until the
first segment is staged, in the stage loops, in the slab epilogues, and the effective shader clock (s_memtime ticks per 100 MHz wall tick).
OPE_WIDE_EXP=1|2|4 (set before the process starts) removes the loop's global loads / MFMAs / LDS deposits: timing experiments only.
    python tools/wide_phases.py            # 3s5z_gall, B = 32"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd.config import default_args
from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_episodes
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
from offpolicy_amd.algorithms.qmix.qmix import QMix
dims = DIMS[os.environ.get("OPE_PHASE_DIMS", "3s5z_gall")]; B = int(os.environ.get("OPE_PHASE_B", "32"))
dev = torch.device("cuda:0"); pinfo = policy_info_for(dims)
args = default_args(gain=1.0, use_soft_update=False)
policy = QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length)
trainer.tune["debug"] = 1
buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, 64, dims.episode_length, True, True, device=dev)
ep = synth_episodes(np.random.RandomState(0), 64, dims, avail="bernoulli")
buf.insert(64, *[{"policy_0": ep[k]} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")])
s = buf.policy_buffers["policy_0"].sample_inds(np.arange(B))
batch = tuple({"policy_0": x} for x in s) + (None, None)
for _ in range(3):
    trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
res = []
for rep in range(5):
    trainer.workspace_view(B, "dbg").zero_()
    trainer.train_policy_on_batch(batch)
    torch.cuda.synchronize()
    d = trainer.workspace_view(B, "dbg").view(torch.int64).cpu().numpy()[:256 * 8].reshape(-1, 8)
    d = d[d[:, 0] > 0]
    res.append(d)
d = res[-1]
cyc, wall = d[:, :4].astype(np.float64), d[:, 4:8].astype(np.float64)
span_wall = (wall[:, 3].max() - wall[:, 0].min()) / 100.0          # us (100 MHz)
clk = (cyc[:, 3] - cyc[:, 0]) / np.maximum(wall[:, 3] - wall[:, 0], 1) * 100.0   # MHz
print("OPE_WIDE_EXP=%s  workgroups %d  kernel span %.1f us  effective clock median %.0f MHz" % (os.environ.get("OPE_WIDE_EXP", "0"), len(d), span_wall, np.median(clk)))
print("per workgroup, cycles (median / max): start->first stage staged %.0f / %.0f   ->end of last stage loop %.0f / %.0f   ->end %.0f / %.0f   total %.0f / %.0f" % (
    np.median(cyc[:, 1] - cyc[:, 0]), (cyc[:, 1] - cyc[:, 0]).max(), np.median(cyc[:, 2] - cyc[:, 1]), (cyc[:, 2] - cyc[:, 1]).max(),
    np.median(cyc[:, 3] - cyc[:, 2]), (cyc[:, 3] - cyc[:, 2]).max(), np.median(cyc[:, 3] - cyc[:, 0]), (cyc[:, 3] - cyc[:, 0]).max()))
print("start skew between workgroups: %.2f us; spans of the 5 repetitions (us): %s" % ((wall[:, 0].max() - wall[:, 0].min()) / 100.0, ", ".join(
    "%.1f" % ((r[:, 7].max() - r[:, 4].min()) / 100.0) for r in res)))
