"""Phase stamps (s_memtime) of workgroup 0 / thread 0 of the MADDPG critic tile kernel (OPE_DDPG_DBG=1).
Usage: OPE_DDPG_DBG=1 python tools/ddpg_phases.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OPE_DDPG_DBG"] = "1"
import bench
from offpolicy_amd import _lib
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
for _ in range(5):
    s = buf.policy_buffers["policy_0"].sample_inds(np.random.choice(4096, B))
    trainer.shared_train_policy_on_batch("policy_0", tuple({"policy_0": x} for x in s) + (None, None))
torch.cuda.synchronize()
v = trainer.workspace_view(B, "fused_slabs")
st = v[-64:].view(torch.int64).cpu().numpy()[:24]
print("staging: issue loads %d, stores %d, issue weights %d, rest %d" % (st[20]-st[0], st[22]-st[20], st[23]-st[22], st[1]-st[23]))
print("agents:", [int(st[8 + i] - (st[1] if i == 0 else st[7 + i])) for i in range(3)])
names = ["stage vectors + inputs", "target actor x N", "target critic fwd", "live critic fwd", "TD", "critic bwd"]
print("cycles:", {n: int(st[i + 1] - st[i]) for i, n in enumerate(names)})
print("total", int(st[6] - st[0]), "s_memtime ticks (~2.4 per ns on this part: 81 k ticks = the launch's ~35 us)")
