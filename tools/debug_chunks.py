"""Compare the chunked two-stream QMIX schedule against the single-stream one (per-tensor gradient differences)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd.config import default_args
from offpolicy_amd.utils.synth import DIMS, EnvDims, policy_info_for, synth_episodes
from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
from offpolicy_amd.algorithms.qmix.qmix import QMix

d0 = DIMS[sys.argv[3] if len(sys.argv) > 3 else "3m"]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dims = EnvDims("x", d0.n_agents, d0.act_dim, d0.obs_dim, d0.state_dim, T)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
args = default_args()
dev = torch.device("cuda:0")
pinfo = policy_info_for(dims)
torch.manual_seed(1)
policy = QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=T)
NE = max(16, B)
buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, NE, T, True, True, device=dev)
ep = synth_episodes(np.random.RandomState(0), NE, dims, avail="bernoulli")
buf.insert(NE, *[{"policy_0": ep[k]} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")])
s = buf.policy_buffers["policy_0"].sample_inds(np.arange(B))
batch = tuple({"policy_0": x} for x in s) + (None, None)
theta0 = trainer.theta.clone(); tgt0 = trainer.theta_tgt.clone()
res = {}
for C in (1, 2, 4, 2, 4, 2):
    os.environ["OPE_CHUNKS"] = str(C)
    trainer.theta.copy_(theta0); trainer.theta_tgt.copy_(tgt0)
    trainer.optimizer.exp_avg.zero_(); trainer.optimizer.exp_avg_sq.zero_(); trainer.optimizer.step_count = 0
    trainer._ws = {}
    info, _, _ = trainer.train_policy_on_batch(batch)
    torch.cuda.synchronize()
    g = trainer.grad.cpu().numpy().copy()
    if C in res:
        print("   repeat C", C, "max diff vs first run of same C", np.abs(g - res[C]).max(), "vs C=1", np.abs(g - res[1]).max())
    else:
        res[C] = g
    print("C", C, "loss", float(info["loss"]), "gnorm", float(info["grad_norm"]), "maxdiff vs C=1", np.abs(g - res[1]).max())
spec = list(policy.q_network.spec().items())
for C in (2, 4):
    print("---- C =", C)
    for name, (shape, off) in spec:
        n = int(np.prod(shape))
        a, b = res[1][off:off + n], res[C][off:off + n]
        print("%-28s max|ref| %.3e  max diff %.3e  nan %d" % (name, np.abs(a).max(), np.abs(a - b).max(), np.isnan(b).sum()))
    a, b = res[1][policy.q_network.padded_numel:], res[C][policy.q_network.padded_numel:]
    print("mixer+tail max diff %.3e" % np.abs(a - b).max())
