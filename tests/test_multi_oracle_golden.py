"""CPU: the several-policies-under-one-mixer oracles (oracle/qmix_oracle.QMixMultiOracle, oracle/mqmix_oracle.MQMixMultiOracle) against
the reference's frozen outputs (tests/golden/*multi*.npz, oracle/make_golden_multi.py): losses, pre-clip gradient norms, Q_tot, priorities,
first-step gradients and every policy's live / target parameters after three steps."""
import numpy as np
import pytest

from conftest import load_golden, sub
from oracle.qmix_oracle import HP, QMixMultiOracle
from oracle.mqmix_oracle import MQMixMultiOracle

M_KEYS = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition",
          "avail_acts", "next_avail_acts")
R_KEYS = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")


@pytest.mark.parametrize("name,mlp", [("mqmix_multi", True), ("mqmix_multi_huber_per", True), ("mvdn_multi_speaker_listener", True),
                                      ("qmix_multi", False), ("qmix_multi_nodouble", False), ("vdn_multi", False)])
def test_multi_policy_oracle_matches_reference(name, mlp):
    g = load_golden(name)
    P = len(g["shapes"])
    vdn = bool(g["vdn"])
    hp = HP(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
            huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]),
            tau=float(g["hp_tau"]), max_grad_norm=float(g["hp_maxnorm"]), use_double_q=bool(g["hp_double_q"]), vdn=vdn)
    agents = [sub(g, "p%d/agent/" % i) for i in range(P)]
    n_total = int(sum(int(r[0]) for r in g["shapes"]))
    orc = (MQMixMultiOracle if mlp else QMixMultiOracle)(agents, None if vdn else sub(g, "mixer/"), n_total, hp)
    keys = M_KEYS if mlp else R_KEYS
    batch = [[g["p%d/batch/%s" % (i, k)] if "p%d/batch/%s" % (i, k) in g else None for k in keys] for i in range(P)]
    w = g["per_weights"] if "per_weights" in g else None
    for s in range(len(g["loss"])):
        out = orc.train_step(batch, weights=w)
        np.testing.assert_allclose(out["loss"], g["loss"][s], rtol=2e-5)
        np.testing.assert_allclose(out["grad_norm"], g["grad_norm"][s], rtol=2e-5)
        np.testing.assert_allclose(out["Q_tot"], g["Q_tot"][s], rtol=2e-5, atol=1e-7)
        if w is not None:
            np.testing.assert_allclose(out["priorities"], g["priorities"][s], rtol=2e-5, atol=1e-7)
        if s == 0:
            coef = min(1.0, hp.max_grad_norm / (float(g["grad_norm"][0]) + 1e-6))
            n_checked = 0
            for k, ref in sub(g, "grad0/").items():          # "p{i}/agent/<name>" or "mixer/<name>"
                key = ("agent/" + k.replace("/agent/", "/", 1)) if k.startswith("p") else k
                got = out["grads"][key]
                np.testing.assert_allclose(got * coef, ref, rtol=0, atol=2e-5 * max(np.abs(ref).max(), 1e-6) + 1e-9, err_msg=k)
                n_checked += 1
            assert n_checked > 10 * P
    for i in range(P):
        for grp, src in (("agent/", orc.agent), ("agent_tgt/", orc.agent_tgt)):
            for k, ref in sub(g, "final/p%d/%s" % (i, grp)).items():
                np.testing.assert_allclose(src["p%d/%s" % (i, k)].numpy(), ref, rtol=0, atol=2e-6, err_msg="p%d %s%s" % (i, grp, k))
    for grp, src in (("mixer/", orc.mixer), ("mixer_tgt/", orc.mixer_tgt)):
        for k, ref in sub(g, "final/" + grp).items():
            np.testing.assert_allclose(src[k].numpy(), ref, rtol=0, atol=2e-6, err_msg=grp + k)
