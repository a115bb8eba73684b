"""CPU-only checks: the C-ABI library loads and exports everything include/ope.h declares; host-side logic
(ring slots, segment trees, parameter layout, init stream) behaves like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, sub


def test_library_exports_every_declared_symbol():
    from offpolicy_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ope.h")).read()
    declared = set(re.findall(r"\b(ope_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"ope_dims", "ope_fields", "ope_qmix_cfg", "ope_adam_cfg"}
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libope.so lacks %s declared in include/ope.h" % name
    assert _lib.lib.ope_version() == 1
    assert _lib.lib.ope_strerror(-1).decode().startswith("invalid")


def test_param_layout_and_episode_bytes_match_survey():
    from offpolicy_amd import _lib
    cfg = _lib.QmixCfg()
    cfg.dims = _lib.Dims(8, 14, 252, 216, 150)
    cfg.batch = 32
    off, siz = (C.c_int64 * 48)(), (C.c_int64 * 48)()
    total = _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz)
    assert sum(siz) == 118791            # SURVEY.md section 8 a6 (51 398 agent + 67 393 mixer)
    assert sum(list(siz)[:22]) == 51398
    assert total >= 118791 and total % 4 == 0 and all(o % 4 == 0 for o in off)
    assert _lib.lib.ope_episode_bytes(C.byref(cfg.dims)) == 1493176
    cfg.dims = _lib.Dims(3, 9, 64, 48, 60)
    assert _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz) > 0 and sum(siz) == 58026
    assert _lib.lib.ope_episode_bytes(C.byref(cfg.dims)) == 73308
    cfg.dims = _lib.Dims(3, 9, 1000, 48, 60)        # unsupported obs width -> error code, no crash
    assert _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz) == -1
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) == -1


def test_weight_gradient_split_count_follows_the_cost_model():
    """The K-split row count of the batched weight-gradient launch is searched per shape (busiest CU's workgroups x rows per split,
    at least three workgroups on the busiest of 256 CUs; ope_api.hip, profiles/r03q_wgrad_rows.txt): 192 rows at 3s5z batch 32
    (28 mixer splits of the 4 800 (t, b) rows), 400 rows with the wide centralized state (12), and the 160-row default for
    launches too small to fill the chip three times (3m: 12 splits of 1 920 rows). Read off the size of the mixer's slab region."""
    import os
    if os.environ.get("OPE_WGRAD_ROWS"):
        pytest.skip("row count pinned by OPE_WGRAD_ROWS")
    from offpolicy_amd import _lib

    def mixer_splits(dims, batch):
        cfg = _lib.QmixCfg()
        cfg.dims, cfg.batch = _lib.Dims(*dims), batch
        off, siz = (C.c_int64 * 48)(), (C.c_int64 * 48)()
        total = _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz)
        n = C.c_int64(0)
        assert _lib.lib.ope_qmix_workspace_find(C.byref(cfg), b"raw_mixer", C.byref(n)) >= 0
        mixer_size = total - off[22]              # padded mixer block of the flat parameter vector
        assert n.value % mixer_size == 0
        return n.value // mixer_size

    assert mixer_splits((8, 14, 252, 216, 150), 32) == 28
    assert mixer_splits((8, 14, 252, 2232, 150), 32) == 12
    assert mixer_splits((3, 9, 64, 48, 60), 32) == 12
    assert mixer_splits((8, 14, 252, 216, 150), 4) == 4       # 600 rows: the default again


@pytest.mark.parametrize("name", ["qmix_shape_hyper1", "qmix_shape_layer2", "qmix_shape_h128"])
def test_non_default_network_shapes_are_refused_loudly(name):
    """The kernels are specialised to the reference's default network shape. The oracle is pinned on three other shapes
    (tests/golden/qmix_shape_*.npz, test_oracle_golden.py); until the engine has a path for them the Python layer must refuse
    them with NotImplementedError naming the option -- never take another path silently."""
    from conftest import load_golden
    from offpolicy_amd.config import default_args, require_reference_architecture
    g = load_golden(name)
    args = default_args(hidden_size=int(g["hp_hidden_size"]), layer_N=int(g["hp_layer_N"]), hypernet_layers=int(g["hp_hypernet_layers"]))
    with pytest.raises(NotImplementedError) as e:
        require_reference_architecture(args)
    assert any(k in str(e.value) for k in ("hidden_size", "layer_N", "hypernet_layers"))
    require_reference_architecture(default_args())      # the defaults pass


def test_second_hidden_block_layout_and_initialisation_match_the_reference():
    """layer_N = 2 (mlp.py:14-28; round 4: supported by the recurrent QMIX / VDN trainer, csrc/ope_block.hip): the C-ABI's parameter
    layout has the reference's 26 agent tensors in named_parameters() order, our constructor consumes the init RNG stream exactly as
    the reference's does (fixture qmix_shape_layer2 = outputs of the real reference), and the configurations the second block cannot
    run are refused by cfg validation."""
    import torch
    from conftest import load_golden
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args, require_reference_architecture
    from offpolicy_amd.algorithms.qmix.algorithm.agent_q_function import init_agent_values, agent_param_names, agent_param_shapes, agent_layout
    from offpolicy_amd.algorithms.qmix.algorithm.q_mixer import init_mixer_values, MIXER_PARAM_NAMES
    g = load_golden("qmix_shape_layer2")
    n, a, d, s, t = [int(x) for x in g["dims"]]
    names = agent_param_names(2)
    assert len(names) == _lib.OPE_QMIX_NPARAM_AGENT_2 and sorted(names) == sorted(k[len("agent/"):] for k in g if k.startswith("agent/"))
    assert [tuple(x) for x in agent_param_shapes(d, a, 2)] == [g["agent/" + k].shape for k in names]
    offs, sizes, total = agent_layout(d, a, 2)
    assert sizes == [int(np.prod(g["agent/" + k].shape)) for k in names]
    assert all(o % 4 == 0 for o in offs) and offs == sorted(offs)
    cfg = _lib.QmixCfg()
    cfg.dims, cfg.batch = _lib.Dims(n, a, d, s, t, 2), 4
    off, siz = (C.c_int64 * 48)(), (C.c_int64 * 48)()
    assert _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz) > 0
    assert list(siz)[26:26 + len(MIXER_PARAM_NAMES)] == [int(np.prod(g["mixer/" + k].shape)) for k in MIXER_PARAM_NAMES]
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) > 0
    torch.manual_seed(1)
    np.random.seed(1)
    av = init_agent_values(d, a, layer_N=2)
    mv = init_mixer_values(n, s)
    for v, k in zip(av, names):
        assert np.array_equal(v.numpy(), g["agent/" + k]), k
    for v, k in zip(mv, MIXER_PARAM_NAMES):
        assert np.array_equal(v.numpy(), g["mixer/" + k]), k
    for bad in (dict(mlp=1), dict(phase=2), dict(time_chunks=2)):
        c2 = _lib.QmixCfg()
        c2.dims, c2.batch = _lib.Dims(n, a, d, s, 1 if "mlp" in bad else t, 2), 4
        for k, v in bad.items():
            setattr(c2, k, v)
        assert _lib.lib.ope_qmix_workspace_bytes(C.byref(c2)) == -1, bad
    c3 = _lib.QmixCfg()
    c3.dims, c3.batch = _lib.Dims(n, a, d, s, t, 3), 4
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(c3)) == -1
    args = default_args(layer_N=2)
    require_reference_architecture(args, allow_layer_N_2=True)
    with pytest.raises(NotImplementedError):
        require_reference_architecture(default_args(layer_N=3), allow_layer_N_2=True)


def test_network_without_input_layernorm_keeps_the_layout_and_hides_two_slots():
    """use_feature_normalization = False (mlp.py:60-62; round 4): the reference's module has no rnn.feature_norm.* parameters. The
    C-ABI's flat layout is unchanged (OPE_DIMS_NO_FEATURE_NORM only tells the first-layer kernels to take the rows as they are); the
    Python module exposes the reference's 20 names, initialises from the same RNG stream as the reference (fixture qmix_shape_nofn =
    outputs of the real reference) and refuses what the flag cannot be combined with."""
    import torch
    from conftest import load_golden
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args, require_reference_architecture
    from offpolicy_amd.algorithms.qmix.algorithm.agent_q_function import init_agent_values, agent_param_names, exposed_agent_names, agent_layout
    g = load_golden("qmix_shape_nofn")
    n, a, d, s, t = [int(x) for x in g["dims"]]
    assert int(g["hp_feature_norm"]) == 0
    exposed = exposed_agent_names(1, feature_norm=False)
    assert exposed == [k[len("agent/"):] for k in g if k.startswith("agent/")] and len(exposed) == 20
    cfg = _lib.QmixCfg()
    cfg.dims, cfg.batch = _lib.Dims(n, a, d, s, t, 1, _lib.OPE_DIMS_NO_FEATURE_NORM), 4
    off, siz = (C.c_int64 * 48)(), (C.c_int64 * 48)()
    total = _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz)
    cfg0 = _lib.QmixCfg()
    cfg0.dims, cfg0.batch = _lib.Dims(n, a, d, s, t), 4
    off0, siz0 = (C.c_int64 * 48)(), (C.c_int64 * 48)()
    assert total == _lib.lib.ope_qmix_param_layout(C.byref(cfg0), off0, siz0) and list(off) == list(off0) and list(siz) == list(siz0)
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) > 0
    torch.manual_seed(1)
    np.random.seed(1)
    av = init_agent_values(d, a)
    for v, k in zip(av, agent_param_names(1)):
        if not k.startswith("rnn.feature_norm."):
            assert np.array_equal(v.numpy(), g["agent/" + k]), k
    cfg.mlp, cfg.dims.episode_length = 1, 1
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) > 0                # round 5: the MLP Q-learning nets too (fixtures mqmix_shape_nofn, ...)
    cfg.mlp, cfg.dims.episode_length, cfg.dims.flags = 0, t, 8
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) == -1              # unknown flag bits
    dd = _lib.DdpgCfg()
    dd.dims, dd.batch, dd.num_q = _lib.Dims(n, a, d, s, 1, 1, _lib.OPE_DIMS_NO_FEATURE_NORM), 4, 1
    assert _lib.lib.ope_ddpg_workspace_bytes(C.byref(dd)) == -1                # the MADDPG families: refused
    args = default_args(use_feature_normalization=False)
    with pytest.raises(NotImplementedError):
        require_reference_architecture(args)
    require_reference_architecture(args, allow_no_feature_norm=True)


def test_tanh_networks_initialise_like_the_reference_and_are_limited_to_what_the_kernels_carry():
    """use_ReLU = False (mlp.py:9-12; round 4, OPE_DIMS_TANH): the orthogonal initialisation uses the tanh gain -- our constructor
    consumes the RNG stream as the reference does (fixture qmix_shape_tanh = outputs of the real reference) -- and the C-ABI accepts
    the flag exactly where trunk_fwd3 / trunk_bwd3 can carry it."""
    import torch
    from conftest import load_golden
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args, require_reference_architecture
    from offpolicy_amd.algorithms.qmix.algorithm.agent_q_function import init_agent_values, agent_param_names
    g = load_golden("qmix_shape_tanh")
    n, a, d, s, t = [int(x) for x in g["dims"]]
    assert int(g["hp_use_relu"]) == 0
    torch.manual_seed(1)
    np.random.seed(1)
    av = init_agent_values(d, a, use_ReLU=False)
    for v, k in zip(av, agent_param_names(1)):
        assert np.array_equal(v.numpy(), g["agent/" + k]), k
    cfg = _lib.QmixCfg()
    cfg.dims, cfg.batch = _lib.Dims(n, a, d, s, t, 1, _lib.OPE_DIMS_TANH), 4
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) > 0
    for bad in (dict(trunk_path=4), dict(phase=2)):
        c2 = _lib.QmixCfg()
        c2.dims, c2.batch = _lib.Dims(n, a, d, s, t, 1, _lib.OPE_DIMS_TANH), 4
        for k, v in bad.items():
            setattr(c2, k, v)
        assert _lib.lib.ope_qmix_workspace_bytes(C.byref(c2)) == -1, bad
    for dims in (_lib.Dims(n, a, 400, s, t, 1, _lib.OPE_DIMS_TANH), _lib.Dims(n, a, d, s, t, 2, _lib.OPE_DIMS_TANH)):
        c3 = _lib.QmixCfg()
        c3.dims, c3.batch = dims, 4
        assert _lib.lib.ope_qmix_workspace_bytes(C.byref(c3)) == -1      # wider than trunk_fwd3's registers / a second hidden block
    args = default_args(use_ReLU=False)
    with pytest.raises(NotImplementedError):
        require_reference_architecture(args)
    require_reference_architecture(args, allow_tanh=True)


def test_action_space_kinds_of_the_maddpg_family_are_validated():
    """Continuous (Box) and multi-discrete action spaces (round 4; MADDPGPolicy.py:73-116): the space stand-ins give upstream's
    dimensions, and the C-ABI checks the combinations it is handed."""
    from offpolicy_amd import _lib
    from offpolicy_amd.utils.spaces import MultiDiscrete, Box, Discrete, get_dim_from_space
    md = MultiDiscrete([[0, 2], [0, 3]])
    assert list(get_dim_from_space(md)) == [3, 4] and get_dim_from_space(Discrete(5)) == 5
    assert get_dim_from_space(Box(low=-np.ones(3, np.float32), high=np.ones(3, np.float32))) == 3

    def cfg_bytes(**kw):
        c = _lib.DdpgCfg()
        c.dims, c.batch, c.num_q = _lib.Dims(2, 7, 10, 12, 1), 6, 1
        heads = kw.pop("heads", None)
        if heads:
            c.n_act_heads = len(heads)
            for i, k in enumerate(heads):
                c.act_head_dims[i] = k
        for k, v in kw.items():
            setattr(c, k, v)
        return _lib.lib.ope_ddpg_workspace_bytes(C.byref(c))
    assert cfg_bytes() > 0 and cfg_bytes(heads=[3, 4]) > 0 and cfg_bytes(continuous=1) > 0
    assert cfg_bytes(heads=[3, 3]) == -1                          # the blocks must add up to act_dim
    assert cfg_bytes(heads=[3, 4], continuous=1) == -1            # one or the other
    assert cfg_bytes(continuous=1, target_gumbel=1) == -1         # continuous target noise is additive, not gumbel
    assert cfg_bytes(continuous=2) == -1
    # the Q-learning families: Box spaces are refused at construction, before anything touches the GPU (upstream asserts a discrete space);
    # MultiDiscrete ones are taken (round 5): the policy exposes upstream's per-head parameter names, the trainer one mixer input per
    # (agent, sub-action); what upstream itself cannot run with them (VDN, the previous action as an input) is refused
    from offpolicy_amd.config import default_args
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy
    pc = lambda sp: {"cent_obs_dim": 9, "cent_act_dim": 14, "obs_space": [6], "share_obs_space": [9], "act_space": sp}
    for P in (QMixPolicy, M_QMixPolicy):
        with pytest.raises(NotImplementedError, match="Discrete / MultiDiscrete action spaces"):
            P({"args": default_args(), "device": "cpu"}, pc(Box(low=-np.ones(3, np.float32), high=np.ones(3, np.float32))))
        pol = P({"args": default_args(), "device": "cpu"}, pc(md))
        assert pol.multidiscrete and pol.head_dims == [3, 4] and pol.output_dim == 7
        names = [k for k, _ in pol.q_network.named_parameters()]
        assert names[-4:] == ["q.action_outs.0.weight", "q.action_outs.0.bias", "q.action_outs.1.weight", "q.action_outs.1.bias"]
        assert [tuple(v.shape) for _, v in pol.q_network.named_parameters()][-4:] == [(3, 64), (3,), (4, 64), (4,)]
        ra = pol.get_random_actions(np.zeros((5, 6), np.float32))
        assert ra.shape == (5, 7) and np.array_equal(ra[:, :3].sum(1), np.ones(5)) and np.array_equal(ra[:, 3:].sum(1), np.ones(5))
    with pytest.raises(NotImplementedError, match="prev_act_inp with a MultiDiscrete"):
        QMixPolicy({"args": default_args(prev_act_inp=True), "device": "cpu"}, pc(md))


def test_struct_mirrors_match_the_compiled_library():
    """ope_abi_sizeof: every ctypes mirror in offpolicy_amd._lib has the size the library was compiled with (also checked at import)."""
    from offpolicy_amd import _lib
    assert len(_lib.ABI_MIRRORS) == 12
    for cname, cls in _lib.ABI_MIRRORS.items():
        assert _lib.lib.ope_abi_sizeof(cname.encode()) == C.sizeof(cls), cname
    assert _lib.lib.ope_abi_sizeof(b"ope_no_such_struct") == -1 and _lib.lib.ope_abi_sizeof(None) == -1


def test_joint_action_of_policies_with_different_action_dimensions_is_validated():
    """ope_rddpg_cfg.joint_act_dim / joint_act_col (round 4; simple_speaker_listener under scripts/train_mpe_rmaddpg.sh: a 3-action speaker and
    a 5-action listener): the critic's first layer is S + joint_act_dim wide whatever the update policy's own act_dim, and the combinations
    that make no sense are refused."""
    from offpolicy_amd import _lib

    def cfg(**kw):
        c = _lib.RddpgCfg()
        c.dims, c.batch, c.num_q = _lib.Dims(1, 3, 3, 14, 5), 4, 1       # the speaker: 1 agent, 3 actions, 3 observations, S = 14
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def critic_fc1_width(c):
        offs, sizes = (C.c_int64 * 22)(), (C.c_int64 * 22)()
        assert _lib.lib.ope_rddpg_param_layout(C.byref(c), 1, offs, sizes) > 0
        return sizes[0]                                                  # feature_norm.weight: one entry per input column
    assert critic_fc1_width(cfg()) == 14 + 3
    assert critic_fc1_width(cfg(joint_act_dim=8, joint_act_col=0)) == 14 + 8
    assert critic_fc1_width(cfg(joint_act_dim=8, joint_act_col=5)) == 14 + 8
    assert _lib.lib.ope_rddpg_workspace_bytes(C.byref(cfg(joint_act_dim=8, joint_act_col=0))) > 0
    assert _lib.lib.ope_rddpg_workspace_bytes(C.byref(cfg(joint_act_dim=8, joint_act_col=6))) == -1       # 6 + 1 * 3 > 8
    assert _lib.lib.ope_rddpg_workspace_bytes(C.byref(cfg(joint_act_dim=8, joint_act_col=-1))) == -1
    assert _lib.lib.ope_rddpg_workspace_bytes(C.byref(cfg(joint_act_dim=8, n_total_agents=2))) == -1      # columns OR equal agent blocks
    assert _lib.lib.ope_rddpg_workspace_bytes(C.byref(cfg(joint_act_dim=2000))) == -1


def test_null_arguments_are_rejected_without_a_gpu():
    from offpolicy_amd import _lib
    assert _lib.lib.ope_adam_step(None, 4, None, None, None, None, None, None, None, None) == -1
    assert _lib.lib.ope_polyak(0, None, None, 0.5, None) == -1
    d = _lib.Dims(2, 5, 12, 10, 6)
    assert _lib.lib.ope_store_gather(C.byref(d), 4, None, None, 2, None, None, None) == -1


@pytest.mark.parametrize("name", ["qmix_tiny", "qmix_3m_katA", "qmix_odd", "qmix_gall_tiny", "qmix_gall_3m"])
def test_initialisation_reproduces_reference_rng_stream(name):
    """torch.manual_seed(1) + our constructors' init draws == the reference modules' initial weights, bit for bit."""
    import torch
    from offpolicy_amd.algorithms.qmix.algorithm.agent_q_function import init_agent_values, AGENT_PARAM_NAMES
    from offpolicy_amd.algorithms.qmix.algorithm.q_mixer import init_mixer_values, MIXER_PARAM_NAMES
    g = load_golden(name)
    n, a, d, s, t = [int(x) for x in g["dims"]]
    torch.manual_seed(1)
    np.random.seed(1)
    av = init_agent_values(d, a, gain_out=float(g["hp_gain"]) if "hp_gain" in g else 0.01)     # --gain: the head's init gain (act.py:10-12)
    mv = init_mixer_values(n, s)
    for v, k in zip(av, AGENT_PARAM_NAMES):
        assert np.array_equal(v.numpy(), g["agent/" + k]), k
    for v, k in zip(mv, MIXER_PARAM_NAMES):
        assert np.array_equal(v.numpy(), g["mixer/" + k]), k


def test_ring_index_matches_reference_fixture():
    from offpolicy_amd.utils.ring import RingIndex
    g = load_golden("qmix_tiny")
    r = RingIndex(6)
    assert np.array_equal(r.next_slots(4), g["pre_idx_range"])
    assert np.array_equal(r.next_slots(5), g["idx_range"])       # wraps: [4, 5, 0, 1, 2]
    assert r.filled_i == int(g["filled_i"]) and r.current_i == int(g["current_i"])
    with pytest.raises(AssertionError):
        r.next_slots(7)


def test_segment_trees_against_bruteforce():
    from offpolicy_amd.utils.segment_tree import SumSegmentTree, MinSegmentTree
    rng = np.random.RandomState(0)
    cap = 64
    st, mt = SumSegmentTree(cap), MinSegmentTree(cap)
    arr = np.zeros(cap)
    arrm = np.full(cap, np.inf)
    for _ in range(30):
        idx = rng.randint(0, cap, size=rng.randint(1, 9))
        val = rng.rand(len(idx)) + 0.01
        st[idx] = val
        mt[idx] = val
        for i, v in zip(idx, val):
            arr[i] = v
            arrm[i] = v
        lo = rng.randint(0, cap)
        hi = rng.randint(lo + 1, cap + 1)
        np.testing.assert_allclose(st.sum(lo, hi), arr[lo:hi].sum(), rtol=1e-12)
        assert mt.min(lo, hi) == arrm[lo:hi].min()
        np.testing.assert_allclose(st.sum(), arr.sum(), rtol=1e-12)
        mass = rng.rand(7) * st.sum()
        got = st.find_prefixsum_idx(mass)
        cs = np.cumsum(arr)
        want = np.searchsorted(cs, mass, side="right")
        assert np.array_equal(got, np.minimum(want, cap - 1))
    with pytest.raises(AssertionError):
        SumSegmentTree(48)


def test_runner_hook_redirects_reference_imports():
    """after install(), the import statements the reference's runners execute resolve to the engine."""
    import importlib
    import sys
    import offpolicy_amd.runner_hook as hook
    before = {k: sys.modules.get(k) for k in hook.MODULE_MAP}
    try:
        done = hook.install(only=("qmix", "rmatd3"))
        assert "offpolicy.algorithms.qmix.qmix" in done and "offpolicy.algorithms.maddpg.maddpg" not in done
        from offpolicy_amd.algorithms.qmix.qmix import QMix
        from offpolicy_amd.utils.rec_buffer import PrioritizedRecReplayBuffer
        assert importlib.import_module("offpolicy.algorithms.qmix.qmix").QMix is QMix
        assert importlib.import_module("offpolicy.utils.rec_buffer").PrioritizedRecReplayBuffer is PrioritizedRecReplayBuffer
        m = importlib.import_module("offpolicy.algorithms.r_matd3.algorithm.rMATD3Policy")
        assert m.R_MATD3Policy.__module__.startswith("offpolicy_amd")
        hook.install()          # everything: every mapped module must import
        for ref_name in hook.MODULE_MAP:
            assert sys.modules[ref_name].__name__.startswith("offpolicy_amd"), ref_name
    finally:
        hook.uninstall()
    for k, v in before.items():
        assert sys.modules.get(k) is v


def test_every_environment_switch_is_documented():
    """Kernels and host code read a few OPE_* environment switches (A/B knobs, defaults = measured best). Each must be named
    in INTEGRATION.md / DESIGN.md so that no hidden mode exists."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "off-policy_amd", "csrc", "*.h*")):
        src = open(f).read()
        names |= set(re.findall(r'getenv\("(OPE_[A-Z0-9_]+)"\)', src))
        names |= set(re.findall(r'env_choice\("(OPE_[A-Z0-9_]+)"', src))
    for f in glob.glob(os.path.join(root, "off-policy_amd", "**", "*.py"), recursive=True) + [os.path.join(root, "bench.py")]:
        names |= set(re.findall(r'environ\.get\("(OPE_[A-Z0-9_]+)"', open(f).read()))
    docs = open(os.path.join(root, "INTEGRATION.md")).read() + open(os.path.join(root, "DESIGN.md")).read()
    assert names, "scan found nothing: pattern out of date"
    missing = sorted(n for n in names if n not in docs)
    assert not missing, "undocumented switches: %s" % missing


def test_one_layer_hyper_networks_layout_and_initialisation_match_the_reference():
    """hypernet_layers = 1 (q_mixer.py:39-44; round 4: supported by the recurrent QMIX trainer on the fused chain kernels): the C-ABI's
    parameter layout has the reference's 10 mixer tensors with its shapes, our constructor consumes the init RNG stream exactly as the
    reference's does (fixture qmix_shape_hyper1 = outputs of the real reference), and the configurations the one-layer form cannot run
    are refused by cfg validation (no silent other path)."""
    import torch
    from offpolicy_amd import _lib
    from offpolicy_amd.algorithms.qmix.algorithm.agent_q_function import init_agent_values, AGENT_PARAM_NAMES
    from offpolicy_amd.algorithms.qmix.algorithm.q_mixer import init_mixer_values, MIXER_PARAM_NAMES_1, mixer_param_shapes
    g = load_golden("qmix_shape_hyper1")
    n, a, d, s, t = [int(x) for x in g["dims"]]
    cfg = _lib.QmixCfg()
    cfg.dims, cfg.batch, cfg.hypernet_layers = _lib.Dims(n, a, d, s, t), 4, 1
    off, siz = (C.c_int64 * 48)(), (C.c_int64 * 48)()
    total = _lib.lib.ope_qmix_param_layout(C.byref(cfg), off, siz)
    assert total > 0
    got = list(siz)[22:32]
    want = [int(np.prod(g["mixer/" + k].shape)) for k in MIXER_PARAM_NAMES_1]
    assert got == want and list(siz)[32] == 0
    assert [tuple(x) for x in mixer_param_shapes(n, s, 1)] == [g["mixer/" + k].shape for k in MIXER_PARAM_NAMES_1]
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) > 0
    torch.manual_seed(1)
    np.random.seed(1)
    av = init_agent_values(d, a)
    mv = init_mixer_values(n, s, hypernet_layers=1)
    for v, k in zip(av, AGENT_PARAM_NAMES):
        assert np.array_equal(v.numpy(), g["agent/" + k]), k
    for v, k in zip(mv, MIXER_PARAM_NAMES_1):
        assert np.array_equal(v.numpy(), g["mixer/" + k]), k
    for bad in (dict(mlp=1), dict(phase=2), dict(mixer_path=3), dict(chain_path=1)):
        c2 = _lib.QmixCfg()
        c2.dims, c2.batch, c2.hypernet_layers = _lib.Dims(n, a, d, s, 1 if "mlp" in bad else t), 4, 1
        for k, v in bad.items():
            setattr(c2, k, v)
        assert _lib.lib.ope_qmix_workspace_bytes(C.byref(c2)) == -1, bad


@pytest.mark.parametrize("name,mlp", [("qmix_md_tiny", False), ("qmix_md_odd_huber_per", False), ("mqmix_md_small", True)])
def test_multi_discrete_policies_reproduce_the_reference_initialisation(name, mlp):
    """MultiDiscrete action spaces for the Q-learning families (round 5): with equal seeds our policy constructor draws the reference's
    initial weights -- one Linear head per sub-action, constructed and re-initialised in order (act.py:14-17) -- exposes them under the
    reference's parameter names in its order, and the mixer is one sized for n_agents x n_heads inputs (qmix.py:49-57). The fixtures hold
    the real reference's initial weights (seeds 1 / 1, policy then trainer)."""
    import torch
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims, policy_info_for
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.mqmix.algorithm.mQMixPolicy import M_QMixPolicy
    from offpolicy_amd.algorithms.qmix.algorithm.q_mixer import init_mixer_values, MIXER_PARAM_NAMES
    g = load_golden(name)
    n, a, d, s, t = [int(x) for x in g["dims"]]
    heads = [int(x) for x in g["multi_discrete"]]
    torch.manual_seed(1)
    np.random.seed(1)
    pol = (M_QMixPolicy if mlp else QMixPolicy)({"args": default_args(), "device": "cpu"}, policy_info_for(EnvDims("fx", n, a, d, s, t), multi_discrete=heads)["policy_0"])
    got = dict(pol.q_network.named_parameters())
    want = sub(g, "agent/")
    assert list(got.keys()) == list(want.keys())
    for k, ref in want.items():
        assert np.array_equal(got[k].detach().numpy(), ref), k
    mv = init_mixer_values(n * len(heads), s)
    for v, k in zip(mv, MIXER_PARAM_NAMES):
        assert np.array_equal(v.numpy(), g["mixer/" + k]), k


@pytest.mark.parametrize("name", ["mqmix_shape_nofn", "mqmix_shape_tanh", "mqmix_var_tanh_nofn_huber_per"])
def test_mlp_q_networks_with_shape_flags_initialise_like_the_reference(name):
    """Round 5: use_feature_normalization = False / use_ReLU = False for the MLP Q-learning family. Same RNG stream as the reference's
    constructor (the tanh gain included), the input LayerNorm's two tensors absent from the exposed parameters, and the C-ABI takes the flags
    in MLP mode (tanh up to the 384 inputs trunk_fwd3 carries)."""
    import torch
    from conftest import load_golden
    from offpolicy_amd import _lib
    from offpolicy_amd.algorithms.mqmix.algorithm.agent_q_function import init_mlp_agent_values, MLP_AGENT_PARAM_NAMES
    g = load_golden(name)
    n, a, d, s, _ = [int(x) for x in g["dims"]]
    relu, fn = bool(g["hp_use_relu"]), "agent/mlp.feature_norm.weight" in g
    torch.manual_seed(1)
    np.random.seed(1)
    vals = init_mlp_agent_values(d, a, use_ReLU=relu)
    for v, k in zip(vals, MLP_AGENT_PARAM_NAMES):
        if fn or not k.startswith("mlp.feature_norm."):
            assert np.array_equal(v.numpy(), g["agent/" + k]), k
    assert [k[len("agent/"):] for k in g if k.startswith("agent/")] == [k for k in MLP_AGENT_PARAM_NAMES if fn or not k.startswith("mlp.feature_norm.")]
    cfg = _lib.QmixCfg()
    flags = (0 if fn else _lib.OPE_DIMS_NO_FEATURE_NORM) | (0 if relu else _lib.OPE_DIMS_TANH)
    cfg.dims, cfg.batch, cfg.mlp = _lib.Dims(n, a, d, s, 1, 1, flags), 6, 1
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) > 0
    cfg.dims.obs_dim = 400
    assert (_lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) > 0) == relu      # tanh: network input <= 384
    cfg.dims.obs_dim, cfg.dims.layer_N = d, 2
    assert _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg)) == -1              # a second hidden block: the recurrent nets only
