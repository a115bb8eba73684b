"""Per-wave phase timing of trunk_fwd4 (ope_trunk4.hip) from its s_memtime stamps (ope_qmix_cfg.debug).
    python tools/trunk4_phases.py        # 3s5z, B = 32"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd.config import default_args
from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_episodes
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
from offpolicy_amd.algorithms.qmix.qmix import QMix
dims = DIMS[os.environ.get("OPE_PHASE_DIMS", "3s5z")]; B = int(os.environ.get("OPE_PHASE_B", "32"))
dev = torch.device("cuda:0"); pinfo = policy_info_for(dims)
policy = QMixPolicy({"args": default_args(), "device": dev}, pinfo["policy_0"])
trainer = QMix(default_args(), dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length)
trainer.tune.update(debug=1, trunk_path=4)
buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, 64, dims.episode_length, True, True, device=dev)
ep = synth_episodes(np.random.RandomState(0), 64, dims, avail="bernoulli")
buf.insert(64, *[{"policy_0": ep[k]} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")])
s = buf.policy_buffers["policy_0"].sample_inds(np.arange(B))
batch = tuple({"policy_0": x} for x in s) + (None, None)
for _ in range(3):
    trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
trainer.workspace_view(B, "dbg").zero_()
trainer.train_policy_on_batch(batch)
torch.cuda.synchronize()
d = trainer.workspace_view(B, "dbg").view(torch.int64).cpu().numpy()[16 * 2400:16 * 2400 + 256 * 12 * 16].reshape(-1, 12, 16)
d = d[d[:, 0, 0] > 0]
print("workgroups", len(d), "tiles per wave min/median/max", d[:, :, 8].min(), np.median(d[:, :, 8]), d[:, :, 8].max(),
      " per SIMD (waves w, w+4, w+8) min/max", (d[:, :4, 8] + d[:, 4:8, 8] + d[:, 8:, 8]).min(), (d[:, :4, 8] + d[:, 4:8, 8] + d[:, 8:, 8]).max())
names = ["weights staged", "rows + LN0", "fc1", "LN1 + saves", "fc2 + LN2", "W_ih + gi stores", "remaining tiles"]
for net in (0, 1):
    x = d[net::2].reshape(-1, 16)
    x = x[(x[:, 8] > 0) & (x[:, 0] > 0)]
    print("net %d (%s): " % (net, "live, saving" if net == 0 else "target") + "  ".join("%s %.0f" % (n, np.median(x[:, k + 1] - x[:, k])) for k, n in enumerate(names)) +
          "   first tile %.0f   total median %.0f max %.0f" % (np.median(x[:, 6] - x[:, 1]), np.median(x[:, 7] - x[:, 0]), (x[:, 7] - x[:, 0]).max()))
t0 = d[:, :, 0].min()
print("kernel span (first start -> last end): %.0f cycles" % (d[:, :, 7].max() - t0))
