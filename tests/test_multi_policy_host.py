"""CPU: host side of the several-policies-under-one-mixer trainers (no kernels run): the flat layout [agent_0 | agent_1 | ... | mixer],
parameter counts, the reference's initial-weight RNG stream, nn.Module views onto the flat vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from test_gpu_multi_policy import build


@pytest.mark.parametrize("name,mlp", [("qmix_multi", False), ("mqmix_multi", True), ("mvdn_multi_speaker_listener", True)])
def test_layout_and_initial_weights_of_multi_policy_trainers(name, mlp):
    g = load_golden(name)
    pids, policies, trainer, batch = build(g, mlp, device="cpu")     # (asserts the initial draws against the reference's inside)
    assert trainer.multi and trainer.policy_ids == pids
    # reference parameter count: every policy's q network + the mixer (qmix.py:66-70)
    n_ref = sum(int(np.prod(v.shape)) for i in range(len(pids)) for v in sub(g, "p%d/agent/" % i).values()) + \
        sum(int(np.prod(v.shape)) for v in sub(g, "mixer/").values())
    assert sum(p.numel() for p in trainer.parameters) == n_ref
    off = 0
    for p in pids:
        q = policies[p].q_network
        assert trainer._poff[p] == off and off % 4 == 0
        assert q._flat.data_ptr() == trainer.theta.data_ptr() + 4 * off          # the policy's network lives inside the trainer's vector
        tq = trainer.target_policies[p].q_network
        assert tq._flat.data_ptr() == trainer.theta_tgt.data_ptr() + 4 * off
        off += q.padded_numel
    assert trainer._mixer_off == off and trainer.numel >= off
    if not trainer.vdn:
        first = min(o for _, o in trainer.mixer.spec().values())
        assert first == off                                                       # mixer block right behind the last policy
        w = trainer.mixer.state_dict()["hyper_w1.0.weight"] if "hyper_w1.0.weight" in trainer.mixer.state_dict() else None
        assert w is None or w.shape[1] == int(g["S"])
    else:
        assert trainer.numel == off
    # a write through the module view is a write into the flat vector
    q0 = policies[pids[-1]].q_network
    name0, (shape0, o0) = next(iter(q0.spec().items()))
    dict(q0.named_parameters())[name0].data.fill_(7.0)
    assert float(trainer.theta[trainer._poff[pids[-1]] + o0]) == 7.0
