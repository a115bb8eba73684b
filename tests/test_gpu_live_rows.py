"""-m gpu: live rows (ope_qmix_cfg.live_rows, off-policy_amd/csrc/ope_live.hip).

The reference pads every sampled episode to episode_length steps and multiplies the Bellman error of every (t, b) with
dones_env[t-1, b] = 1 by zero (offpolicy/algorithms/qmix/qmix.py:161-166; normaliser :184-186, priorities :177-181, Q_tot :198). The
engine finds those rows on the device at the start of every step and runs its kernels on the packed remainder. Checked here:
  * the plan kernel against a numpy restatement, on monotone, non-monotone, all-dead, never-ending and tied patterns;
  * a step on live rows against the SAME step on every padded row (same trainer state, same batch) -- with the workspace poisoned with
    NaN in between, so any read of a row the live step did not write shows --, for every option the chain kernels carry;
  * the reference's own fixtures, with the live path pinned;
  * several steps on changing batches (stale rows of earlier steps in the workspace).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from gpu_util import build_from_fixture, batch_from

pytestmark = pytest.mark.gpu
LIVE_LAUNCHES = ("live_plan", "trunk_fwd4_live<", "mixer_hyp_live<", "gru_fwd4_live<", "qchain", "gru_bwd4_live<", "trunk_bwd4_live", "wgrad2_live<")


def plan_reference(dones_env, N):
    """numpy restatement of live_plan_kernel (LivePlan, ope_common.h). dones_env: [T, B]."""
    T, B = dones_env.shape
    lens = np.ones(B, np.int64)
    for b in range(B):
        nz = np.nonzero(dones_env[:, b] != 1.0)[0]
        lens[b] = 1 if len(nz) == 0 else nz[-1] + 2
    perm = sorted(range(B), key=lambda b: (-lens[b], b))
    inv = np.empty(B, np.int64)
    inv[perm] = np.arange(B)
    ls = lens[perm]
    nn = np.array([(ls > t).sum() for t in range(T + 2)], np.int64)
    cum = np.concatenate(([0], np.cumsum(nn)))[:T + 2]
    RL, R1L, TBL = N * cum[T + 1], N * cum[T], cum[T]
    srcrow = -np.ones(RL, np.int64)
    prevrow = -np.ones(RL, np.int64)
    for t in range(T + 1):
        for a in range(N):
            for b in range(B):
                if t < lens[b]:
                    j = inv[b]
                    p = N * cum[t] + a * nn[t] + j
                    srcrow[p] = (t * N + a) * B + b
                    prevrow[p] = N * cum[t - 1] + a * nn[t - 1] + j if t > 0 else -1
    tbrec = np.zeros((TBL, 8), np.int64)
    tbsrc = np.zeros(TBL, np.int64)
    for t in range(T):
        for b in range(B):
            if t < lens[b]:
                j = inv[b]
                q = cum[t] + j
                tbrec[q] = [t, b, N * cum[t] + j, nn[t], N * cum[t + 1] + j, nn[t + 1], int(j < nn[t + 1]), j]
                tbsrc[q] = t * B + b
    return dict(hdr=np.array([RL, R1L, TBL, ls[0]]), len=ls, perm=np.array(perm), cum=cum, nn=nn, srcrow=srcrow, prevrow=prevrow, tbrec=tbrec, tbsrc=tbsrc)


def _patterns(T, B, rng):
    t = np.arange(T)[:, None]
    L = rng.randint(1, T + 1, size=B)
    mono = (t >= (L[None, :] - 1)).astype(np.float32)
    out = {"monotone": mono, "all_dead": np.ones((T, B), np.float32), "never_ends": np.zeros((T, B), np.float32)}
    tied = mono.copy()
    tied[:, 1::2] = tied[:, :1]                      # equal lengths: the ranking must keep batch order among them
    out["ties"] = tied
    holes = mono.copy()                              # flags that go back to 0 after a 1, fractional flags (the mask is 1 - dones_env)
    holes[rng.randint(0, T, size=B), np.arange(B)] = 0.0
    holes[rng.randint(0, T, size=B), np.arange(B)] = 0.5
    out["holes"] = holes
    return out


@pytest.mark.parametrize("T,N,B", [(12, 3, 7), (150, 8, 32), (5, 2, 1), (60, 3, 200)])
def test_plan_kernel_matches_numpy(T, N, B):
    from offpolicy_amd import _lib
    cfg = _lib.QmixCfg()
    cfg.dims = _lib.Dims(N, 6, 64, 48, T, 1, 0)
    cfg.batch, cfg.use_double_q, cfg.gamma = B, 1, 0.99
    need = _lib.lib.ope_qmix_workspace_bytes(C.byref(cfg))
    assert need > 0
    ws = torch.empty(int(need), dtype=torch.uint8, device="cuda")
    ws.fill_(0xFF)
    _lib.check(_lib.lib.ope_qmix_workspace_init(C.byref(cfg), _lib.ptr(ws), ws.numel(), _lib.current_stream()), "init")
    n = C.c_int64(0)
    off = _lib.lib.ope_qmix_workspace_find(C.byref(cfg), b"live_plan", C.byref(n))
    assert off >= 0
    plan = ws[off:off + 4 * n.value].view(torch.int32)
    steps = 0
    for name, de in _patterns(T, B, np.random.RandomState(T + B)).items():
        d = torch.from_numpy(de).cuda().contiguous()
        _lib.check(_lib.lib.ope_qmix_live_plan(C.byref(cfg), _lib.ptr(d), _lib.ptr(ws), ws.numel(), _lib.current_stream()), "plan")
        steps += 1
        torch.cuda.synchronize()
        got = plan.cpu().numpy().astype(np.int64)
        ref = plan_reference(de, N)
        r4 = lambda x: (x + 3) & ~3
        o = 16
        np.testing.assert_array_equal(got[:4], ref["hdr"], err_msg=name)
        np.testing.assert_array_equal(got[o:o + B], ref["len"], err_msg=name); o += r4(B)
        np.testing.assert_array_equal(got[o:o + B], ref["perm"], err_msg=name); o += r4(B)
        np.testing.assert_array_equal(got[o:o + T + 2], ref["cum"], err_msg=name); o += r4(T + 2)
        np.testing.assert_array_equal(got[o:o + T + 2], ref["nn"], err_msg=name); o += r4(T + 2)
        RL, R1L, TBL = ref["hdr"][:3]
        np.testing.assert_array_equal(got[o:o + 8 * TBL].reshape(TBL, 8), ref["tbrec"], err_msg=name); o += 8 * T * B
        np.testing.assert_array_equal(got[o:o + TBL], ref["tbsrc"], err_msg=name); o += r4(T * B)
        np.testing.assert_array_equal(got[o:o + RL], ref["srcrow"], err_msg=name); o += r4((T + 1) * N * B)
        np.testing.assert_array_equal(got[o:o + RL], ref["prevrow"], err_msg=name)
        acc = plan[8:16].view(torch.int64).cpu().numpy()
        assert acc[3] == steps and acc[0] >= RL
        # the zero-filled regions
        for reg in (b"err_abs", b"loss_part"):
            m = C.c_int64(0)
            ro = _lib.lib.ope_qmix_workspace_find(C.byref(cfg), reg, C.byref(m))
            assert not ws[ro:ro + 4 * m.value].view(torch.float32).any()


# ---- a step on live rows against the same step on every padded row --------------------------------------------------------------
def _make(dims, args, B, seed, vdn=False, dones=None, avail="bernoulli"):
    from offpolicy_amd.utils.synth import synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    torch.manual_seed(seed)
    np.random.seed(seed)
    dev = torch.device("cuda:0")
    pinfo = policy_info_for(dims)
    policy = QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length, vdn=vdn)
    # a target net that differs from the live one, as after some training
    trainer.theta_tgt.add_(0.01 * torch.randn_like(trainer.theta_tgt))
    n_ep = 3 * B
    buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, n_ep, dims.episode_length, True, True, device=dev)
    ep = synth_episodes(np.random.RandomState(seed), n_ep, dims, avail=avail)
    if dones is not None:
        ep["dones_env"] = dones(ep["dones_env"])
        ep["dones"] = np.repeat(ep["dones_env"][:, :, None, :], dims.n_agents, axis=2)
    d = as_policy_dicts(ep)
    buf.insert(n_ep, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    return policy, trainer, buf


def _poison(trainer, B):
    from offpolicy_amd import _lib
    ws = trainer._ws[B]
    ws.fill_(0xFF)                      # 0xFFFFFFFF is a NaN as float32 and -1 as an index
    cfg = trainer._cfg(B)
    _lib.check(_lib.lib.ope_qmix_workspace_init(C.byref(cfg), _lib.ptr(ws), ws.numel(), _lib.current_stream()), "ope_qmix_workspace_init")


def _one(trainer, batch, live):
    from offpolicy_amd import _lib
    trainer.tune["live_rows"] = 2 if live else 1
    info, prio, _ = trainer.train_policy_on_batch(batch)
    launched = ",".join(_lib.last_launches())
    torch.cuda.synchronize()
    return {k: float(v) for k, v in info.items()}, (None if prio is None else np.asarray(prio.cpu() if torch.is_tensor(prio) else prio).copy()), \
        trainer.grad[:trainer.numel + 4].clone(), launched


def _snapshot(trainer):
    o = trainer.optimizer
    return trainer.theta.clone(), trainer.theta_tgt.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step_count


def _restore(trainer, s):
    o = trainer.optimizer
    trainer.theta.copy_(s[0]); trainer.theta_tgt.copy_(s[1]); o.exp_avg.copy_(s[2]); o.exp_avg_sq.copy_(s[3]); o.step_count = s[4]


def _compare(a, b, trainer, tag, gtol=2e-5):
    """(info, priorities, gradient) of two runs of the same step: same row-wise arithmetic, different summation order over rows."""
    for k in a[0]:
        np.testing.assert_allclose(b[0][k], a[0][k], rtol=2e-5, atol=1e-7, err_msg="%s %s" % (tag, k))
    if a[1] is not None:
        np.testing.assert_allclose(b[1], a[1], rtol=1e-5, atol=1e-7, err_msg=tag + " priorities")
    ga, gb = a[2].cpu().numpy(), b[2].cpu().numpy()
    assert np.isfinite(gb).all(), tag
    assert ga[trainer.numel + 1] == gb[trainer.numel + 1], (tag, "mask count")
    worst = 0.0
    for name, p in _segments(trainer):
        x, y = ga[p], gb[p]
        scale = max(np.abs(x).max(), 1e-12)
        err = np.abs(x - y).max() / scale
        worst = max(worst, err)
        assert err <= gtol, (tag, name, err)
    return worst


def _segments(trainer):
    pol = trainer.policies["policy_0"]
    for name, (shape, off) in pol.q_network.spec().items():
        yield "agent/" + name, slice(off, off + int(np.prod(shape)))
    if not trainer.vdn:
        for name, (shape, off) in trainer.mixer.spec().items():
            yield "mixer/" + name, slice(off, off + int(np.prod(shape)))


def _holes(de):
    de = de.copy()
    rng = np.random.RandomState(5)
    T, E = de.shape[:2]
    de[rng.randint(0, T, size=E), np.arange(E), 0] = 0.0
    de[:, 0] = 1.0           # an episode with no live step but the first
    de[:, 1] = 0.0           # one that never ends: all T + 1 agent rows
    return de


CONFIGS = {
    # name: (dims, B, args overrides, vdn, dones transform)
    "3m": (("3m",), 8, {}, False, None),
    "3m_vdn": (("3m",), 8, {}, True, None),
    "3m_huber_per": (("3m",), 8, dict(use_huber_loss=True, huber_delta=0.5, use_per=True), False, None),
    "3m_nodouble": (("3m",), 6, dict(use_double_q=False), False, None),
    "3m_hyper1": (("3m",), 8, dict(hypernet_layers=1), False, None),
    "3m_holes": (("3m",), 9, {}, False, _holes),
    "3m_nofn": (("3m",), 5, dict(use_feature_normalization=False), False, None),
    "d124_n5": ((5, 6, 124, 100, 20), 7, {}, False, None),
    "d188_a20": ((3, 20, 188, 40, 9), 5, {}, False, _holes),             # two head tiles
    "d370": ((4, 7, 370, 322, 11), 6, {}, False, None),                  # the 24-chunk trunk, 8-byte rows; S % 4 != 0: 8-byte state rows
    "d252_n10": ((10, 9, 252, 216, 7), 4, {}, False, _holes),            # two agents per wave of the chain kernel (pinned chain_path = 2)
}


def _dims_of(spec):
    from offpolicy_amd.utils.synth import DIMS, EnvDims
    return DIMS[spec[0]] if len(spec) == 1 else EnvDims("custom", *spec)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_live_step_equals_padded_step(name):
    from offpolicy_amd.config import default_args
    spec, B, over, vdn, dones = CONFIGS[name]
    dims = _dims_of(spec)
    args = default_args(**over)
    policy, trainer, buf = _make(dims, args, B, seed=3, vdn=vdn, dones=dones)
    trainer.tune.update(trunk_path=4, chain_path=2, wgrad_path=2, scan_family=4)
    snap = _snapshot(trainer)
    rng = np.random.RandomState(11)
    worst = 0.0
    for step in range(3):                # changing batches: rows of earlier steps stay in the workspace
        inds = rng.choice(len(buf), B, replace=False)
        w = rng.uniform(0.2, 1.0, size=B).astype(np.float32) if args.use_per else None
        batch = batch_from(buf, inds, w)
        s0 = _snapshot(trainer)
        ref = _one(trainer, batch, live=False)
        assert "_live" not in ref[3] and "live_plan" not in ref[3], ref[3]
        s1 = _snapshot(trainer)
        _restore(trainer, s0)
        _poison(trainer, B)
        got = _one(trainer, batch, live=True)
        for want in LIVE_LAUNCHES:
            assert want in got[3] or (vdn and want.startswith("mixer_hyp")), (want, got[3])      # (VDN: no hyper-networks)
        assert "_live" in [x for x in got[3].split(",") if x.startswith("qchain")][0], got[3]
        worst = max(worst, _compare(ref, got, trainer, "%s step %d" % (name, step)))
        # the optimizer saw the same gradient: parameters agree too
        np.testing.assert_allclose(trainer.theta.cpu().numpy(), s1[0].cpu().numpy(), rtol=0, atol=2e-6)
        trainer.soft_target_updates()
    from golden_util import record_errors
    record_errors("live_vs_padded:" + name, {"worst_grad": float(worst)})
    _restore(trainer, snap)


def test_full_length_episodes_take_every_row():
    """Episodes that never end: the plan is the identity ranking, every one of the (T + 1) N B rows is live."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import DIMS
    dims, B = DIMS["3m"], 8
    policy, trainer, buf = _make(dims, default_args(), B, seed=4, dones=lambda de: np.zeros_like(de))
    trainer.tune.update(trunk_path=4, chain_path=2, wgrad_path=2, scan_family=4)
    batch = batch_from(buf, np.arange(B))
    s0 = _snapshot(trainer)
    ref = _one(trainer, batch, live=False)
    _restore(trainer, s0)
    _poison(trainer, B)
    got = _one(trainer, batch, live=True)
    _compare(ref, got, trainer, "full length")
    hdr = trainer.workspace_view(B, "live_plan").view(torch.int32)[:4].cpu().numpy()
    T, N = dims.episode_length, dims.n_agents
    assert list(hdr) == [(T + 1) * N * B, T * N * B, T * B, T + 1]


def test_live_rows_that_cannot_run_are_an_error_or_a_fallback():
    """live_rows = 2 on a configuration the packed kernels do not take (D = 12: no trunk_fwd4) fails before the first launch; "by shape"
    (0) quietly computes every padded row there."""
    from offpolicy_amd import _lib
    g = load_golden("qmix_tiny")
    dims, buf, policy, trainer = build_from_fixture(g)
    batch = batch_from(buf, g["inds"])
    trainer.tune["live_rows"] = 2
    with pytest.raises(_lib.OpeError):
        trainer.train_policy_on_batch(batch)
    assert _lib.last_launches() == []
    trainer.tune["live_rows"] = 0
    info, _, _ = trainer.train_policy_on_batch(batch)
    assert "live_plan" not in _lib.last_launches()
    np.testing.assert_allclose(float(info["loss"]), g["loss"][0], rtol=1e-4)


# ---- the reference's own fixtures on the live path ------------------------------------------------------------------------------------
REF_CASES = ["qmix_3m_katA", "qmix_gall_3m", "qmix_var_d124", "qmix_var_d188", "qmix_var_d252", "qmix_var_d370", "qmix_var_nofn_d252"]


@pytest.mark.parametrize("name", REF_CASES)
def test_reference_fixtures_on_live_rows(name):
    from offpolicy_amd import _lib
    from test_gpu_qmix import _flat_named, RTOL, GRAD_TOL
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune.update(trunk_path=4, chain_path=2, wgrad_path=2, scan_family=4, live_rows=2)
    soft = bool(g["hp_soft_update"]) if "hp_soft_update" in g else True
    hard_after = set(int(x) for x in g["hard_update_after"]) if "hard_update_after" in g else set()
    batch = batch_from(buf, g["inds"])
    for s in range(len(g["loss"])):
        info, _, _ = trainer.train_policy_on_batch(batch)
        launched = ",".join(_lib.last_launches())
        for want in LIVE_LAUNCHES:
            assert want in launched, (want, launched)
        if s == 0:
            cnt = float(trainer.grad[trainer.numel + 1])
            coef = min(1.0, float(g["hp_maxnorm"]) / (float(g["grad_norm"][0]) + 1e-6))
            got = _flat_named(trainer, trainer.grad[:trainer.numel] * (coef / cnt))
            for k, ref in sub(g, "grad0/").items():
                np.testing.assert_allclose(got[k], ref, rtol=0, atol=GRAD_TOL * max(np.abs(ref).max(), 1e-6), err_msg="grad " + k)
        if soft:
            trainer.soft_target_updates()
        elif s in hard_after:
            trainer.hard_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["Q_tot"]), g["Q_tot"][s], rtol=RTOL, atol=1e-6)
    live = _flat_named(trainer, trainer.theta)
    for grp, key in (("final_agent/", "agent/"), ("final_mixer/", "mixer/")):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(live[key + k], ref, rtol=0, atol=3e-5, err_msg=grp + k)


# ---- the plan built outside the step, from the store's flags (ope_store_gather_attach_live / ope_store_live_plan) --------------------------
def _plan_tables_equal(got, ref, T, N, B):
    RL, R1L, TBL = [int(x) for x in ref[:3]]
    r4 = lambda x: (x + 3) & ~3
    assert torch.equal(got[:8], ref[:8])
    o = 16      # every table up to the entries the plan defines (beyond the live counts the regions keep whatever was there)
    for cnt, size in ((B, r4(B)), (B, r4(B)), (T + 2, r4(T + 2)), (T + 2, r4(T + 2)), (8 * TBL, 8 * T * B), (TBL, r4(T * B)), (RL, r4((T + 1) * N * B)), (RL, r4((T + 1) * N * B))):
        assert torch.equal(got[o:o + cnt], ref[o:o + cnt]), o
        o += size


def test_plan_built_by_the_gather_launch_is_the_plan_kernels():
    """RecPolicyBuffer.sample_inds(live_for=trainer): a few extra workgroups of the gather launch build the step's plan from the STORE's flags
    of the sampled episodes. Same tables and row maps as live_plan_kernel builds from the batch; the step that follows skips its own plan
    launch and produces the same gradient and priorities BIT FOR BIT (PER: the per-row error array is cleared inside the step); a tag that is
    not the latest one (another batch sampled since) is not trusted. ope_store_live_plan: the same plan as a launch of its own."""
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import DIMS
    dims, B = DIMS["3m"], 8
    T, N = dims.episode_length, dims.n_agents
    policy, trainer, buf = _make(dims, default_args(use_per=True), B, seed=6, dones=_holes)
    trainer.tune.update(trunk_path=4, chain_path=2, wgrad_path=2, scan_family=4)
    pb = buf.policy_buffers["policy_0"]
    inds = np.array([3, 0, 17, 9, 1, 22, 5, 5])
    w = np.linspace(0.3, 1.0, B).astype(np.float32)
    snap = _snapshot(trainer)
    # (a) plain: the step launches the plan kernel
    s = pb.sample_inds(inds)
    assert getattr(s[5], "_ope_live", None) is None
    ref = _one(trainer, tuple({"policy_0": x} for x in s) + (w, inds), live=True)
    assert "live_plan" in ref[3].split(","), ref[3]
    plan_ref = trainer.workspace_view(B, "live_plan").view(torch.int32).clone()
    # (b) built by the gather launch, twice in a row (the second time over the first one's plan)
    trainer.tune["live_rows"] = 0
    for rep in range(2):
        _restore(trainer, snap)
        if rep == 0:
            _poison(trainer, B)
        s = pb.sample_inds(inds, live_for=trainer)
        assert s[5]._ope_live is not None
        for x, y in zip(s, pb.sample_inds(inds)):      # the batch itself is what a plain gather returns
            assert torch.equal(x, y)
        info, prio, _ = trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s) + (w, inds))
        launched = _lib.last_launches()
        torch.cuda.synchronize()
        assert "live_plan" not in launched and "trunk_fwd4_live<4>" in launched and "qchain_live<1,1>" in launched, launched
        _plan_tables_equal(trainer.workspace_view(B, "live_plan").view(torch.int32).clone(), plan_ref, T, N, B)
        assert torch.equal(trainer.grad[:trainer.numel + 4], ref[2])
        np.testing.assert_array_equal(np.asarray(prio), ref[1])
    # (c) a tag that is not the latest: the step builds its own plan again
    _restore(trainer, snap)
    s1 = pb.sample_inds(inds, live_for=trainer)
    s2 = pb.sample_inds(inds[::-1].copy(), live_for=trainer)
    info, _, _ = trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s1) + (w, inds))
    assert "live_plan" in _lib.last_launches()
    assert torch.equal(trainer.grad[:trainer.numel + 4], ref[2])
    # (d) the same plan as a launch of its own, into the second region (host indices in the kernel arguments, and device indices)
    cfg = trainer._cfg(B)
    ws = trainer._ws[B]
    tgt = _lib.LiveTarget()
    _lib.check(_lib.lib.ope_qmix_live_target(C.byref(cfg), _lib.ptr(ws), ws.numel(), 1, C.byref(tgt)), "target")
    for dev_inds in (False, True):
        trainer.workspace_view(B, "live_plan1").view(torch.int32)[:8].zero_()
        hi = np.ascontiguousarray(inds, dtype=np.int64)
        di = torch.from_numpy(hi).cuda()
        _lib.check(_lib.lib.ope_store_live_plan(pb.buffer_size, T, _lib.ptr(pb.dones_env), _lib.ptr(di) if dev_inds else None,
                                                None if dev_inds else hi.ctypes.data_as(C.c_void_p), C.byref(tgt), _lib.current_stream()), "ope_store_live_plan")
        torch.cuda.synchronize()
        _plan_tables_equal(trainer.workspace_view(B, "live_plan1").view(torch.int32).clone(), plan_ref, T, N, B)
    # (e) a trainer pinned to every padded row gets no plan
    trainer.tune["live_rows"] = 1
    assert pb.sample_inds(inds, live_for=trainer)[5]._ope_live is None


def test_live_only_gather_moves_the_live_entries_and_nothing_else():
    """RecPolicyBuffer.sample_inds(live_for=trainer, live_only=True): of obs and share_obs the gather launch writes only the time entries
    t < len_b (len_b as the plan defines it -- non-monotone flags included, _holes). Checked: (i) into a destination filled with NaN, every
    live entry equals the plain gather's bit for bit and every dead entry is still NaN (nothing else was moved), the short-row fields are
    whole; (ii) the step on that batch -- NaN in every row it must not read -- gives the plain batch's gradient and priorities BIT FOR BIT;
    (iii) a step that would not run on this plan's live rows refuses the batch."""
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims
    dims, B = EnvDims("custom", 3, 9, 188, 216, 24), 8      # obs and state rows long enough for the episode-contiguous step path (>= 32 16-byte pieces)
    T, N = dims.episode_length, dims.n_agents
    policy, trainer, buf = _make(dims, default_args(use_per=True), B, seed=6, dones=_holes)
    trainer.tune.update(trunk_path=4, chain_path=2, wgrad_path=2, scan_family=4)
    pb = buf.policy_buffers["policy_0"]
    inds = np.array([3, 0, 17, 9, 1, 22, 5, 5])
    w = np.linspace(0.3, 1.0, B).astype(np.float32)
    snap = _snapshot(trainer)
    plain = pb.sample_inds(inds)
    ref = _one(trainer, tuple({"policy_0": x} for x in plain) + (w, inds), live=True)
    trainer.tune["live_rows"] = 0
    de = plain[5][:, :, 0].cpu().numpy()      # [T, B]
    lens = plan_reference(de, N)
    len_b = np.empty(B, np.int64)
    len_b[lens["perm"]] = lens["len"]
    assert len_b.min() < T, "the case needs dead entries"
    _restore(trainer, snap)
    out = pb.alloc_batch(B)
    for v in out.values():
        if torch.is_tensor(v):
            v.fill_(float("nan"))
    s = pb.sample_inds(inds, live_for=trainer, live_only=True, out=out)
    assert s[5]._ope_live is not None and s[5]._ope_live[4] is True
    torch.cuda.synchronize()
    obs, share = s[0].cpu().numpy(), s[1].cpu().numpy()              # [N, T+1, B, D], [T+1, B, S]
    pobs, pshare = plain[0].cpu().numpy(), plain[1].cpu().numpy()
    for b in range(B):
        L = int(len_b[b])
        np.testing.assert_array_equal(obs[:, :L, b], pobs[:, :L, b])
        np.testing.assert_array_equal(share[:L + 1, b], pshare[:L + 1, b])      # (the target mixer of the last live step reads state entry len_b: times zero)
        assert np.isnan(obs[:, L:, b]).all() and np.isnan(share[L + 1:, b]).all(), b
    for k in (2, 3, 4, 5, 6):      # acts, rewards, dones, dones_env, avail_acts: whole
        assert torch.equal(s[k], plain[k]), k
    info, prio, _ = trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s) + (w, inds))
    launched = _lib.last_launches()
    torch.cuda.synchronize()
    assert "live_plan" not in launched and any(k.startswith("trunk_fwd4_live<") for k in launched), launched
    assert torch.equal(trainer.grad[:trainer.numel + 4], ref[2])
    np.testing.assert_array_equal(np.asarray(prio), ref[1])
    assert np.isfinite(float(info["loss"]))
    # (iii) the tag is not the latest one any more: a step that would build its own plan over every row of this batch must not run
    _restore(trainer, snap)
    s1 = pb.sample_inds(inds, live_for=trainer, live_only=True)
    pb.sample_inds(inds[::-1].copy(), live_for=trainer)
    with pytest.raises(RuntimeError, match="live_only"):
        trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s1) + (w, inds))
    # a trainer pinned to every padded row gets a whole batch whatever was asked
    trainer.tune["live_rows"] = 1
    s2 = pb.sample_inds(inds, live_for=trainer, live_only=True)
    assert s2[5]._ope_live is None
    for x, y in zip(s2, plain):
        assert torch.equal(x, y)


@pytest.mark.parametrize("at", [0, 1, 3])
def test_batches_gathered_ahead_on_a_side_stream_train_the_same(at):
    """RecPolicyBuffer.sample_inds_ahead (+ midstep_event / ope_qmix_signal_event): four consecutive updates with the gather of step k + 1
    launched on the side stream -- behind everything enqueued so far (at = 0) or from launch `at` of step k on -- against the same four updates
    with sample_inds in front of every step: the parameters after the last step are equal BIT FOR BIT (same batches, same plans -- in
    alternating plan regions --, same arithmetic), no step launches a plan kernel of its own, and the handles' arrays are the plain gather's."""
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims
    dims, B = EnvDims("custom", 3, 9, 188, 216, 24), 8
    policy, trainer, buf = _make(dims, default_args(), B, seed=11, dones=_holes)
    trainer.tune.update(trunk_path=4, chain_path=2, wgrad_path=2, scan_family=4)
    pb = buf.policy_buffers["policy_0"]
    rng = np.random.RandomState(3)
    inds = [rng.randint(0, 3 * B, size=B) for _ in range(4)]
    snap = _snapshot(trainer)
    for ix in inds:      # sequential
        s = pb.sample_inds(ix, live_for=trainer, live_only=True)
        trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s) + (None, None))
        trainer.soft_target_updates()
    torch.cuda.synchronize()
    ref = trainer.theta.clone(), trainer.theta_tgt.clone()
    _restore(trainer, snap)
    cur = pb.sample_inds_ahead(inds[0], live_for=trainer, live_only=True)
    for k in range(4):
        s = cur.get()
        if k == 0:
            for x, y in zip(s, pb.sample_inds(inds[0])):      # the live entries are the plain gather's (dead ones: whatever the slot held)
                if x.shape == y.shape and x.dim() == 3 and x.shape[-1] == 1:
                    assert torch.equal(x, y)
        mid = pb.midstep_event(at) if at else None
        trainer.train_policy_on_batch(tuple({"policy_0": x} for x in s) + (None, None))
        assert "live_plan" not in _lib.last_launches(), _lib.last_launches()
        if k + 1 < 4:
            cur = pb.sample_inds_ahead(inds[k + 1], live_for=trainer, live_only=True, after=mid)
        trainer.soft_target_updates()
    torch.cuda.synchronize()
    assert torch.equal(trainer.theta, ref[0]) and torch.equal(trainer.theta_tgt, ref[1])


@pytest.mark.parametrize("name", ["3m", "3m_holes", "3m_nofn", "d370", "3m_huber_per"])
def test_finalize_folded_into_the_slab_sum_gives_the_same_gradient(name):
    """ope_set_w2_fin: wgrad2 + w2_fin (the LayerNorm-fed-Linear identities applied per workgroup slab, one launch sums the slabs into the flat
    gradient and leaves the clip norm's partials) against wgrad2 + w2_reduce + finalize -- same gradient to rounding, same loss tail, the same
    parameters after the optimizer step (the norm partials of either path feed ope_adam_step), on live rows and on every padded row; and a
    step of the other path afterwards is not disturbed by the partials this one left (the workspace is shared)."""
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args
    spec, B, over, vdn, dones = CONFIGS[name]
    dims = _dims_of(spec)
    policy, trainer, buf = _make(dims, default_args(**over), B, seed=4, vdn=vdn, dones=dones)
    trainer.tune.update(trunk_path=4, chain_path=2, wgrad_path=2, scan_family=4)
    pb = buf.policy_buffers["policy_0"]
    inds = np.arange(B)
    w = np.linspace(0.3, 1.0, B).astype(np.float32) if over.get("use_per") else None
    batch = tuple({"policy_0": x} for x in pb.sample_inds(inds)) + (w, inds if w is not None else None)
    snap = _snapshot(trainer)
    try:
        for live in (True, False):
            res = {}
            for fin in (0, 1, 0):
                _lib.lib.ope_set_w2_fin(fin)
                _restore(trainer, snap)
                out = _one(trainer, batch, live)
                assert ("w2_fin<" in out[3]) == bool(fin), out[3]
                res.setdefault(fin, []).append((out, trainer.theta.clone()))
            a, b, a2 = res[0][0], res[1][0], res[0][1]
            _compare(a[0], b[0], trainer, "%s live=%s fin" % (name, live), gtol=2e-5)
            assert (a[1] - b[1]).abs().max() <= 2e-6 * max(float(a[1].abs().max()), 1.0), "parameters after the step"
            assert torch.equal(a[0][2], a2[0][2]) and torch.equal(a[1], a2[1]), "the unfused path after a fused step on the same workspace"
    finally:
        _lib.lib.ope_set_w2_fin(1)
