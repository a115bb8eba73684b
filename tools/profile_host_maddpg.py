"""Host-side cost of the eager MADDPG-MLP step (sample + train + soft update), cProfile over 300 steps:
    python tools/profile_host_maddpg.py"""
import cProfile, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from offpolicy_amd.config import default_args
from offpolicy_amd.utils.synth import DIMS, policy_info_for
from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
from offpolicy_amd.algorithms.maddpg.algorithm.MADDPGPolicy import MADDPGPolicy
from offpolicy_amd.algorithms.maddpg.maddpg import MADDPG
dims = DIMS["simple_spread"]; dev = torch.device("cuda:0"); B = 256
pinfo = policy_info_for(dims)
policy = MADDPGPolicy({"args": default_args(), "device": dev}, pinfo["policy_0"])
trainer = MADDPG(default_args(), dims.n_agents, {"policy_0": policy}, lambda x: "policy_0", device=dev)
trainer.device_noise = True
buf = MlpReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, 4096, True, True, False, device=dev)
tr = bench.ddpg_transitions(np.random.RandomState(0), 4096, dims)
buf.insert(4096, *[{"policy_0": tr[k]} for k in bench.DDPG_KEYS])


def step():
    sample = buf.sample(B)
    trainer.train_policy_on_batch("policy_0", sample)
    policy.soft_target_updates()


for _ in range(50):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host loop %.1f us/step, with final sync %.1f us/step" % (1e6 * (t1 - t0) / 300, 1e6 * (t2 - t0) / 300))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
