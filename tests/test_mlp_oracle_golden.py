"""Pin oracle/mqmix_oracle.py against outputs of the REAL reference (tests/golden/mqmix_*.npz, mvdn_*.npz)."""
import numpy as np
import pytest

from conftest import load_golden, sub
from oracle import mqmix_oracle as MO
from oracle.qmix_oracle import HP

CASES = ["mqmix_spread", "mqmix_small_huber_per", "mqmix_small_nodouble", "mvdn_spread",
         # round 5: use_feature_normalization = False / use_ReLU = False (tanh) for the MLP family (oracle/make_golden_mlp.py, OPE_GOLDEN_ONLY=shapes)
         "mqmix_shape_nofn", "mqmix_shape_tanh", "mqmix_var_tanh_nofn_huber_per", "mvdn_var_tanh_nofn"]
T_KEYS = ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones", "dones_env", "valid_transition",
          "avail_acts", "next_avail_acts")


def mlp_store_from(g):
    """Replay the fixture's inserts into reference-layout transition arrays (mlp_buffer.py:125-155 defaults)."""
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    cap = int(g["cap"])
    st = dict(obs=np.zeros((cap, n, d), np.float32), share_obs=np.zeros((cap, s), np.float32), acts=np.zeros((cap, n, a), np.float32),
              rewards=np.zeros((cap, n, 1), np.float32), next_obs=np.zeros((cap, n, d), np.float32),
              next_share_obs=np.zeros((cap, s), np.float32), dones=np.ones((cap, n, 1), np.float32),
              dones_env=np.ones((cap, 1), np.float32), valid_transition=np.zeros((cap, n, 1), np.float32),
              avail_acts=np.ones((cap, n, a), np.float32), next_avail_acts=np.ones((cap, n, a), np.float32))
    if "pre_idx_range" in g:
        for k in T_KEYS:
            st[k][g["pre_idx_range"]] = g["pre_tr/" + k]
    for k in T_KEYS:
        st[k][g["idx_range"]] = g["tr/" + k]
    return st


def mlp_oracle_from(g):
    hp = HP(gamma=float(g["hp_gamma"]), lr=float(g["hp_lr"]), opti_eps=float(g["hp_eps"]), use_huber_loss=bool(g["hp_huber"]),
            huber_delta=float(g["hp_delta"]), use_per=bool(g["hp_per"]), per_nu=float(g["hp_nu"]), per_eps=float(g["hp_per_eps"]),
            tau=float(g["hp_tau"]), max_grad_norm=float(g["hp_maxnorm"]), use_double_q=bool(g["hp_double_q"]), vdn=bool(g["vdn"]))
    hp.use_relu = bool(g["hp_use_relu"]) if "hp_use_relu" in g else True
    return MO.MQMixOracle(sub(g, "agent/"), sub(g, "mixer/") if not bool(g["vdn"]) else None, int(g["dims"][0]), hp)


@pytest.mark.parametrize("name", CASES)
def test_sample_and_train_match_reference(name):
    g = load_golden(name)
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    assert list(sub(g, "agent/").keys()) == list(MO.mlp_agent_param_shapes(d, a, feature_norm="agent/mlp.feature_norm.weight" in g).keys())
    store = mlp_store_from(g)
    batch = MO.sample_inds(store, g["inds"])
    for k, got in zip(T_KEYS, batch):
        assert np.array_equal(got, g["batch/" + k]), k
    orc = mlp_oracle_from(g)
    w = g["per_weights"] if "per_weights" in g else None
    for st in range(len(g["loss"])):
        out = orc.train_step(batch, weights=w)
        np.testing.assert_allclose(out["loss"], g["loss"][st], rtol=2e-5)
        np.testing.assert_allclose(out["grad_norm"], g["grad_norm"][st], rtol=2e-5)
        np.testing.assert_allclose(out["Q_tot"], g["Q_tot"][st], rtol=2e-5, atol=1e-7)
        if w is not None:
            np.testing.assert_allclose(out["priorities"], g["priorities"][st], rtol=2e-5, atol=1e-7)
    for grp, dst in (("final_agent/", orc.agent), ("final_agent_tgt/", orc.agent_tgt), ("final_mixer/", orc.mixer),
                     ("final_mixer_tgt/", orc.mixer_tgt)):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(dst[k].numpy(), ref, rtol=0, atol=2e-5, err_msg=grp + k)


@pytest.mark.parametrize("name", ["mqmix_md_small", "mqmix_md_huber_per"])
def test_multi_discrete_train_steps_match_reference(name):
    """MultiDiscrete action spaces under MLP QMIX (mqmix.py:41-51, 116-130, 144-155; mQMixPolicy.py:47-55): one q head per sub-action, chosen /
    greedy / target q per head, one mixer input per (agent, sub-action). Oracle only -- the engine's Q-learning policies refuse these spaces
    (DESIGN.md section 2). Double-Q only and no availability masks: upstream's other branches fail on the list of heads."""
    g = load_golden(name)
    orc = mlp_oracle_from(g)
    heads = [int(x) for x in g["multi_discrete"]]
    assert MO.q_head_dims(orc.agent) == heads
    assert orc.mixer["hyper_w1.2.weight"].shape[0] == int(g["dims"][0]) * len(heads) * 32
    batch = tuple(g["batch/" + k] if "batch/" + k in g else None for k in T_KEYS)
    assert batch[9] is None and batch[10] is None
    w = g["per_weights"] if "per_weights" in g else None
    for st in range(len(g["loss"])):
        out = orc.train_step(batch, weights=w)
        np.testing.assert_allclose(out["loss"], g["loss"][st], rtol=2e-5)
        np.testing.assert_allclose(out["grad_norm"], g["grad_norm"][st], rtol=2e-5)
        np.testing.assert_allclose(out["Q_tot"], g["Q_tot"][st], rtol=2e-5, atol=1e-7)
        if w is not None:
            np.testing.assert_allclose(out["priorities"], g["priorities"][st], rtol=2e-5, atol=1e-7)
    for grp, dst in (("final_agent/", orc.agent), ("final_agent_tgt/", orc.agent_tgt), ("final_mixer/", orc.mixer),
                     ("final_mixer_tgt/", orc.mixer_tgt)):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(dst[k].numpy(), ref, rtol=0, atol=2e-5, err_msg=grp + k)


def test_mlp_init_reproduces_reference_rng_stream():
    import torch
    from offpolicy_amd.algorithms.mqmix.algorithm.agent_q_function import init_mlp_agent_values, MLP_AGENT_PARAM_NAMES
    from offpolicy_amd.algorithms.qmix.algorithm.q_mixer import init_mixer_values, MIXER_PARAM_NAMES
    g = load_golden("mqmix_spread")
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    torch.manual_seed(1)
    np.random.seed(1)
    for v, k in zip(init_mlp_agent_values(d, a), MLP_AGENT_PARAM_NAMES):
        assert np.array_equal(v.numpy(), g["agent/" + k]), k
    for v, k in zip(init_mixer_values(n, s), MIXER_PARAM_NAMES):
        assert np.array_equal(v.numpy(), g["mixer/" + k]), k
