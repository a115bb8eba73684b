"""-m gpu: the HIP QMIX/VDN training step (through the C-ABI) against the reference's frozen outputs and the oracle.

Tolerances (fp32, different reduction orders): loss / grad_norm / Q_tot rtol 1e-4,
priorities rtol 1e-4, gradients 5e-5 of the tensor's max magnitude (GRAD_TOL: ~10x the worst measured, also for the pinned-kernel tests since
round 6), parameters after 3 Adam steps atol 3e-5.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from golden_util import oracle_from, reference_store_from
from gpu_util import build_from_fixture, batch_from

pytestmark = pytest.mark.gpu
CASES = ["qmix_tiny", "qmix_tiny_huber_per", "qmix_tiny_nodouble", "vdn_tiny", "qmix_odd", "qmix_3m_katA", "qmix_tiny_prevact", "qmix_tiny_pershare",
         "qmix_gall_tiny", "qmix_gall_3m", "qmix_gall_odd",   # gall: scripts/train_smac_qmix.sh (wide state, gain 1, hard target updates)
         # round 4: shapes that reach the kernel variants bench.py runs when the kernels are pinned (tests below); here: what "by shape" picks
         "qmix_var_d252", "qmix_var_d188", "qmix_var_d124", "qmix_var_d370", "qmix_var_mix216", "qmix_var_mix100", "qmix_var_s2232",
         "qmix_var_n10", "qmix_var_a20",       # 10 agents (two per wave of the fused chain kernel), 20 actions (two head tiles; plain-max targets)
         # one-layer hyper-networks (--hypernet_layers 1, q_mixer.py:39-44): the tiny shape fixture, 8 agents at S = 216, odd S + Huber + PER
         "qmix_shape_hyper1", "qmix_var_hyper1_mix", "qmix_var_hyper1_odd", "qmix_shape_layer2",
         "qmix_var_layer2_d252", "qmix_var_layer2_hyper1", "qmix_var_layer2_odd", "vdn_var_layer2",
         # no input LayerNorm (--use_feature_normalization off, mlp.py:60-62): tiny, the 3s5z width, odd + prev-act + Huber + PER, with a second
         # block and one-layer hyper-networks, VDN
         "qmix_shape_nofn", "qmix_var_nofn_d252", "qmix_var_nofn_odd", "qmix_var_nofn_layer2_hyper1", "vdn_var_nofn",
         # tanh in the agent network's MLP base (--use_ReLU off, mlp.py:9-12): trunk_fwd3 / trunk_bwd3 carry it
         "qmix_shape_tanh", "qmix_var_tanh_d252", "qmix_var_tanh_odd", "vdn_var_tanh",
         # round 5: MultiDiscrete action spaces (one q head per sub-action, one mixer input per (agent, sub-action): QMixPolicy.py:76-93,
         # qmix.py:49-57): two and three heads, Huber + PER weights, plain (non double-Q) per-head targets
         "qmix_md_tiny", "qmix_md_odd_huber_per", "qmix_md_nodouble"]
RTOL = 1e-4
GRAD_TOL = 5e-5      # of the tensor's max magnitude: 10x the worst error any of the 40 fixtures shows on the GPU (5.0e-6, qmix_var_s2232; most: 6e-7 .. 1.2e-6; profiles/r05_parity_errors.txt)


def _flat_named(trainer, flat):
    """{group/name: np.ndarray} views of a flat vector using the trainer's modules' specs."""
    out = {}
    pol = trainer.policies["policy_0"]
    for name, (shape, off) in pol.q_network.spec().items():
        n = int(np.prod(shape))
        out["agent/" + name] = flat[off:off + n].view(shape).cpu().numpy()
    if not trainer.vdn:
        for name, (shape, off) in trainer.mixer.spec().items():
            n = int(np.prod(shape))
            out["mixer/" + name] = flat[off:off + n].view(shape).cpu().numpy()
    return out


@pytest.mark.parametrize("name", CASES)
def test_train_steps_match_reference(name):
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    w = g["per_weights"] if "per_weights" in g else None
    batch = batch_from(buf, g["inds"], w)
    soft = bool(g["hp_soft_update"]) if "hp_soft_update" in g else True
    hard_after = set(int(x) for x in g["hard_update_after"]) if "hard_update_after" in g else set()
    for s in range(len(g["loss"])):
        info, prio, _ = trainer.train_policy_on_batch(batch)
        if s == 0:
            cnt = float(trainer.grad[trainer.numel + 1])
            coef = min(1.0, float(g["hp_maxnorm"]) / (float(g["grad_norm"][0]) + 1e-6))
            got = _flat_named(trainer, trainer.grad[:trainer.numel] * (coef / cnt))
            worst = {}
            for k, ref in sub(g, "grad0/").items():
                tol = GRAD_TOL * max(np.abs(ref).max(), 1e-6)
                worst[k] = float(np.abs(got[k] - ref).max() / max(np.abs(ref).max(), 1e-6))
                np.testing.assert_allclose(got[k], ref, rtol=0, atol=tol, err_msg="grad " + k)
            from golden_util import record_errors
            record_errors("qmix_grad0:" + name, {"worst": max(worst.values()), "tensor": max(worst, key=worst.get)})
            for k in got:
                if ".fc_h." in k:
                    assert not np.any(got[k]), k
        if soft:
            trainer.soft_target_updates()
        elif s in hard_after:                 # --use_soft_update given (= off): the runner's hard copy, base_runner.py:281-284
            trainer.hard_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["Q_tot"]), g["Q_tot"][s], rtol=RTOL, atol=1e-6)
        if w is not None:
            np.testing.assert_allclose(prio, g["priorities"][s], rtol=RTOL, atol=1e-6)
    live, tgt = _flat_named(trainer, trainer.theta), _flat_named(trainer, trainer.theta_tgt)
    for grp, src, key in (("final_agent/", live, "agent/"), ("final_agent_tgt/", tgt, "agent/"),
                          ("final_mixer/", live, "mixer/"), ("final_mixer_tgt/", tgt, "mixer/")):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(src[key + k], ref, rtol=0, atol=3e-5, err_msg=grp + k)
    # the nn.Module views see the same memory (checkpoint path of the reference runner)
    sd = policy.q_network.state_dict()
    assert list(sd.keys()) == list(sub(g, "agent/").keys())
    head = "q.action_out.weight" if "q.action_out.weight" in sd else "q.action_outs.1.weight"      # (MultiDiscrete: upstream's per-head names)
    np.testing.assert_array_equal(sd[head].cpu().numpy(), live["agent/" + head])


@pytest.mark.parametrize("family,waves", [(4, 4), (4, 2), (1, 0)])
@pytest.mark.parametrize("name", ["qmix_tiny", "qmix_odd"])
def test_every_scan_kernel_family_matches_reference(name, family, waves):
    """The GRU scans have three kernel shapes chosen by row count (ope_gru4.hip with 4 or 2 compute waves per row,
    ope_gru1.hip); the fixtures are small, so pin each shape in turn and repeat the reference comparison."""
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune.update(scan_family=family, scan_waves=waves)      # per trainer (ope_qmix_cfg.scan_family / scan_waves), not process-wide
    batch = batch_from(buf, g["inds"])
    for s in range(len(g["loss"])):
        info, _, _ = trainer.train_policy_on_batch(batch)
        trainer.soft_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
    live = _flat_named(trainer, trainer.theta)
    for k, ref in sub(g, "final_agent/").items():
        np.testing.assert_allclose(live["agent/" + k], ref, rtol=0, atol=3e-5, err_msg=k)


# (fixture, trunk_path) pairs that CAN run: path 3 takes every width; path 4 (trunk_fwd4 / trunk_bwd4) needs an input width that is a
# multiple of 4 with ceil(D / 16) in {4, 8, 12, 16}: D = 64 (3m), 124, 188, 252 (the 3s5z width: KCM 16 with a 12-float tail chunk) -- or an
# even width with 24 chunks: D = 370, the MMM2 width (rows 8-byte aligned only, a 2-float tail, two W_ih tiles read from L2)
TRUNK_CASES = [(n, 3) for n in ("qmix_tiny", "qmix_odd", "qmix_3m_katA", "qmix_tiny_prevact", "qmix_gall_3m", "qmix_tiny_huber_per", "qmix_var_d252", "qmix_var_d370",
                                 "qmix_var_nofn_d252")] + \
              [(n, 4) for n in ("qmix_3m_katA", "qmix_gall_3m", "qmix_var_d124", "qmix_var_d188", "qmix_var_d252", "qmix_var_d370", "qmix_var_nofn_d252")]


def _expect_trunk(dims, path):
    if path == 4:
        return ["trunk_fwd4<%d>" % ((dims.obs_dim + 15) // 16), "trunk_bwd4"]
    return ["trunk_fwd3<", "trunk_bwd3"]


@pytest.mark.parametrize("name,path", TRUNK_CASES)
def test_every_forward_trunk_kernel_matches_reference(name, path):
    """The forward trunk of the two agent nets has two kernel families chosen by shape (ope_qmix_cfg.trunk_path: 3 = trunk_fwd3, one
    launch per net, weights in registers; 4 = trunk_fwd4, both nets in one launch, weights in LDS, a wave per 16-row tile; the trunk
    adjoint follows: trunk_bwd3 / trunk_bwd4): pin each on fixtures whose shape it can run -- path 3 on odd widths (D = 18: 8-byte
    pieces; 12 + 5: 1-float tail), partial tiles, the ring / PER / Huber variants; path 4 on every K-chunk count it is instantiated for
    (KCM 4, 8, 12, 16 = D 64, 124, 188, 252, the last with the 12-float tail chunk of the 3s5z width) -- compare with the reference,
    saved activations included (they feed the backward pass: gradients and final parameters), and ASSERT from the launch log that the
    pinned kernels are the ones that ran (a request the shape does not allow is an error, not a fall-back: next test)."""
    from offpolicy_amd import _lib
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune["trunk_path"] = path
    w = g["per_weights"] if "per_weights" in g else None
    soft = bool(g["hp_soft_update"]) if "hp_soft_update" in g else True
    hard_after = set(int(x) for x in g["hard_update_after"]) if "hard_update_after" in g else set()
    batch = batch_from(buf, g["inds"], w)
    for s in range(len(g["loss"])):
        info, _, _ = trainer.train_policy_on_batch(batch)
        # (with the LDS-resident trunk pinned, "by shape" may run the step on live rows -- the `_live` instantiations of the same kernels,
        # round 6 -- where the rest of the configuration allows it: the family pinned here is what is asserted)
        launched = ",".join(_lib.last_launches()).replace("_live", "")
        in_dim = dims.obs_dim + (dims.act_dim if getattr(policy, "prev_act_inp", False) else 0)
        for want in _expect_trunk(dims._replace(obs_dim=in_dim), path):
            assert want in launched, (want, launched)
        if s == 0:
            cnt = float(trainer.grad[trainer.numel + 1])
            coef = min(1.0, float(g["hp_maxnorm"]) / (float(g["grad_norm"][0]) + 1e-6))
            got = _flat_named(trainer, trainer.grad[:trainer.numel] * (coef / cnt))
            from golden_util import assert_grads_close
            assert_grads_close("qmix_pinned:" + name, got, sub(g, "grad0/"), GRAD_TOL, floor=1e-6)
        if soft:
            trainer.soft_target_updates()
        elif s in hard_after:
            trainer.hard_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
    live = _flat_named(trainer, trainer.theta)
    for k, ref in sub(g, "final_agent/").items():
        np.testing.assert_allclose(live["agent/" + k], ref, rtol=0, atol=3e-5, err_msg=k)


@pytest.mark.parametrize("name,knob,value", [("qmix_tiny", "trunk_path", 4), ("qmix_odd", "trunk_path", 4), ("qmix_odd", "mixer_path", 1),
                                             ("qmix_gall_3m", "mixer_path", 1), ("qmix_tiny", "mixer_path", 1)])
def test_pinned_kernel_that_cannot_run_the_shape_is_an_error(name, knob, value):
    """An explicit ope_qmix_cfg.trunk_path / mixer_path that the shape does not allow (trunk_fwd4 on D = 12 / 18; the resident-weight
    mixer on S = 54 / 10 -- not multiples of 4 -- or S = 240 > 224) returns OPE_EINVAL: no silent fall-back to another kernel, so a
    test that pins a kernel cannot pass on a different one (VERDICT r3 "what's weak" 1)."""
    from offpolicy_amd import _lib
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune[knob] = value
    trainer.tune["chain_path"] = 1
    with pytest.raises(_lib.OpeError):
        trainer.train_policy_on_batch(batch_from(buf, g["inds"]))


# (fixture, mixer_path) pairs that CAN run: 1 = mixer_fwd3 (S % 4 == 0, S <= 224, N <= 8), 2 = mixer_fwd2 (any), 3 = the wide-state GEMM (any)
MIXER_CASES = [(n, 1) for n in ("qmix_3m_katA", "qmix_var_mix216", "qmix_var_mix100", "qmix_var_d124")] + \
              [(n, 2) for n in ("qmix_gall_tiny", "qmix_gall_3m", "qmix_gall_odd", "qmix_tiny", "qmix_odd", "qmix_3m_katA", "qmix_var_mix216")] + \
              [(n, 3) for n in ("qmix_gall_tiny", "qmix_gall_3m", "qmix_gall_odd", "qmix_tiny", "qmix_odd", "qmix_3m_katA", "qmix_var_s2232")]


@pytest.mark.parametrize("name,path", MIXER_CASES)
def test_every_forward_mixer_kernel_matches_reference(name, path):
    """The forward mixer has three kernels chosen by shape (ope_qmix_cfg.mixer_path: 1 weights resident in registers, 2 weights
    streamed per 16-row workgroup, 3 the wide-state form -- first hyper-layers as one stream-K GEMM + second stage from its partial
    slabs, ope_mixer_wide.hip); the fixtures are small, so pin each in turn on shapes it can run and repeat the reference comparison:
    path 1 at the 3s5z state width with eight agents (S = 216, N = 8: the <14, FULL> instantiation bench.py runs) and at narrower ones
    (S = 100, 48, 24: the guarded K loop, fewer agents than waves); path 2 / 3 on narrow, wide (--use_global_all_local_state: S = 34, 83
    -- nothing a multiple of 4 --, 240) and 3m states; path 3 also at the real S = 2 232 (70 K stages, a 24-float tail, two 128-row
    blocks whose K ranges are cut across workgroups). The launch log must show the pinned kernel."""
    from offpolicy_amd import _lib
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune["mixer_path"] = path
    trainer.tune["chain_path"] = 1           # the forward mixer as a kernel of its own (the fused chain replaces paths 1 / 2: next test)
    soft = bool(g["hp_soft_update"]) if "hp_soft_update" in g else True
    hard_after = set(int(x) for x in g["hard_update_after"]) if "hard_update_after" in g else set()
    batch = batch_from(buf, g["inds"])
    want = {1: "mixer_fwd3<14,%d>" % int((dims.state_dim + 15) // 16 == 14), 2: "mixer_fwd2<", 3: "mixer_wide_gemm<"}[path]
    for s in range(len(g["loss"])):
        info, _, _ = trainer.train_policy_on_batch(batch)
        launched = ",".join(_lib.last_launches())
        assert want in launched and (path != 3 or "mixer_fwd2_wide<" in launched), (want, launched)
        if s == 0:
            cnt = float(trainer.grad[trainer.numel + 1])
            coef = min(1.0, float(g["hp_maxnorm"]) / (float(g["grad_norm"][0]) + 1e-6))
            got = _flat_named(trainer, trainer.grad[:trainer.numel] * (coef / cnt))
            from golden_util import assert_grads_close
            assert_grads_close("qmix_pinned:" + name, got, sub(g, "grad0/"), GRAD_TOL, floor=1e-6)
        if soft:
            trainer.soft_target_updates()
        elif s in hard_after:
            trainer.hard_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["Q_tot"]), g["Q_tot"][s], rtol=RTOL, atol=1e-6)
    live, tgt = _flat_named(trainer, trainer.theta), _flat_named(trainer, trainer.theta_tgt)
    for grp, src, key in (("final_agent/", live, "agent/"), ("final_mixer/", live, "mixer/"), ("final_mixer_tgt/", tgt, "mixer/")):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(src[key + k], ref, rtol=0, atol=3e-5, err_msg=grp + k)


CHAIN_CASES = ["qmix_tiny", "qmix_tiny_huber_per", "qmix_tiny_nodouble", "vdn_tiny", "qmix_odd", "qmix_3m_katA", "qmix_tiny_prevact", "qmix_tiny_pershare",
               "qmix_var_mix216", "qmix_var_n10", "qmix_var_a20", "qmix_var_d252"]


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("name", CHAIN_CASES)
def test_both_forms_of_the_row_chain_match_reference(name, path):
    """Between the GRU scan and its adjoint the step runs the (t, b)-row chain -- q heads, chosen / double-Q target selection, both mixers,
    TD target + mask + loss (+ PER weights), the mixer's and the head's adjoints -- either as four launches (ope_qmix_cfg.chain_path = 1:
    head_fwd, mixer_fwd, mixer_bwd, head_bwd) or as two (2: mixer_hyp = the mixers' first hyper-layers + qchain = everything else,
    ope_chain.hip; what "by shape" picks). Pin each form and repeat the reference comparison on MSE / Huber + PER weights, plain-max targets,
    VDN, odd sizes, 3m, previous-action inputs, per-agent centralized observations, 8 agents at the 3s5z state width, 10 agents (two per wave
    of the fused kernel) with 18 actions and 20 actions (two 16-action head tiles); the launch log must show the pinned form."""
    from offpolicy_amd import _lib
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune["chain_path"] = path
    w = g["per_weights"] if "per_weights" in g else None
    batch = batch_from(buf, g["inds"], w)
    for s in range(len(g["loss"])):
        info, prio, _ = trainer.train_policy_on_batch(batch)
        launched = ";".join(_lib.last_launches())
        if path == 2:
            assert ("qchain_vdn<" if bool(g["vdn"]) else "qchain<") in launched and "head_fwd" not in launched and "mixer_fwd" not in launched, launched
            assert bool(g["vdn"]) or "mixer_hyp<" in launched, launched
        else:
            assert "qchain" not in launched and "head_fwd" in launched and "head_bwd_rows" in launched, launched
        if s == 0:
            cnt = float(trainer.grad[trainer.numel + 1])
            coef = min(1.0, float(g["hp_maxnorm"]) / (float(g["grad_norm"][0]) + 1e-6))
            got = _flat_named(trainer, trainer.grad[:trainer.numel] * (coef / cnt))
            from golden_util import assert_grads_close
            assert_grads_close("qmix_pinned:" + name, got, sub(g, "grad0/"), GRAD_TOL, floor=1e-6)
        trainer.soft_target_updates()
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["Q_tot"]), g["Q_tot"][s], rtol=RTOL, atol=1e-6)
        if w is not None:
            np.testing.assert_allclose(prio, g["priorities"][s], rtol=RTOL, atol=1e-6)
    live, tgt = _flat_named(trainer, trainer.theta), _flat_named(trainer, trainer.theta_tgt)
    for grp, src, key in (("final_agent/", live, "agent/"), ("final_agent_tgt/", tgt, "agent/"), ("final_mixer/", live, "mixer/"), ("final_mixer_tgt/", tgt, "mixer/")):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(src[key + k], ref, rtol=0, atol=3e-5, err_msg=grp + k)


@pytest.mark.parametrize("name", ["qmix_var_d252", "qmix_var_d188", "qmix_var_d124"])
def test_observation_rows_read_in_place_from_the_store_give_the_gathered_step(name):
    """Lazy batch: with RecPolicyBuffer.lazy_obs the observation rows stay in the episode-major
    store and trunk_fwd4 / wgrad fetch them through the sampled episode slots (ope_qmix_loss_and_grad_ref). Same arithmetic on the same
    values in the same order: every step's gradient vector and the parameters after it are BIT-identical to the gathered step's, the
    reference fixture's losses hold, and the launch log shows the row-reading variants actually ran. (Reference: rec_buffer.py:206-238
    + qmix.py:108-109.)"""
    from offpolicy_amd import _lib
    from offpolicy_amd.utils.rec_buffer import StoreObs
    g = load_golden(name)
    runs = []
    for lazy in (False, True):
        dims, buf, policy, trainer = build_from_fixture(g)
        trainer.tune["trunk_path"] = 4          # (56 rows: "by shape" would pick the register-resident trunk, which reads a gathered batch only)
        trainer.tune["wgrad_path"] = 1          # (the row-reading form exists for the tile-per-wave kernel: compare it with THAT kernel on gathered rows)
        pb = buf.policy_buffers["policy_0"]
        pb.lazy_obs = lazy
        assert trainer.obs_ref_ok(len(g["inds"]))
        grads, losses = [], []
        for s in range(len(g["loss"])):
            smp = pb.sample_inds(g["inds"])
            assert isinstance(smp[0], StoreObs) == lazy
            info, _, _ = trainer.train_policy_on_batch(tuple({"policy_0": x} for x in smp) + (None, None))
            launched = _lib.last_launches()
            kc = (dims.obs_dim + 15) // 16
            want = ("trunk_fwd4_store<%d>" % kc, "wgrad_store<4>") if lazy else ("trunk_fwd4<%d>" % kc, "wgrad<4>")
            assert all(w in launched for w in want), launched
            trainer.soft_target_updates()
            grads.append(trainer.grad.clone())
            losses.append(float(info["loss"]))
        np.testing.assert_allclose(losses, g["loss"], rtol=RTOL)
        runs.append((grads, trainer.theta.clone(), trainer.theta_tgt.clone()))
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


def test_trainer_gathers_the_rows_itself_where_the_kernels_cannot_read_the_store():
    """A StoreObs handed to a configuration whose first-layer kernels read a gathered batch only (odd obs_dim; previous action as an
    input) is materialised by the trainer: same results as the gathered batch, no error, no other path silently."""
    from offpolicy_amd.utils.rec_buffer import StoreObs
    for name in ("qmix_odd", "qmix_tiny_prevact"):
        g = load_golden(name)
        dims, buf, policy, trainer = build_from_fixture(g)
        pb = buf.policy_buffers["policy_0"]
        pb.lazy_obs = True
        assert not trainer.obs_ref_ok(len(g["inds"])) or policy.prev_act_inp
        for s in range(len(g["loss"])):
            smp = pb.sample_inds(g["inds"])
            assert isinstance(smp[0], StoreObs)
            info, _, _ = trainer.train_policy_on_batch(tuple({"policy_0": x} for x in smp) + (None, None))
            trainer.soft_target_updates()
            np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
            np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)


def test_full_size_step_with_observations_read_from_the_store_is_bit_identical():
    """BASELINE config 4 (3s5z, B = 32) as bench.py runs it by default: observations left in a 48-episode store, repeated indices,
    kernels picked by shape. The gradient of the row-reading step equals the gathered step's bit for bit (which the full-size test
    above compares with the oracle element by element)."""
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import DIMS, synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    dims, nb, cap = DIMS["3s5z"], 32, 48
    args = default_args()
    torch.manual_seed(1)
    np.random.seed(1)
    dev = torch.device("cuda:0")
    policy = QMixPolicy({"args": args, "device": dev}, policy_info_for(dims)["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length)
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, True, True, device=dev)
    for n in (24, 24):
        d = as_policy_dicts(synth_episodes(np.random.RandomState(n), n, dims, avail="bernoulli"))
        buf.insert(n, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    pb = buf.policy_buffers["policy_0"]
    inds = np.random.RandomState(3).randint(0, cap, size=nb)
    inds[5] = inds[4]
    assert trainer.obs_ref_ok(nb)
    trainer.tune["wgrad_path"] = 1          # (the row-reading form belongs to the tile-per-wave kernel; "by shape" takes the register-blocked one on gathered rows)
    snap = (trainer.theta.clone(), trainer.theta_tgt.clone())
    out = []
    for lazy in (False, True):
        trainer.theta.copy_(snap[0]); trainer.theta_tgt.copy_(snap[1])
        trainer.optimizer.exp_avg.zero_(); trainer.optimizer.exp_avg_sq.zero_(); trainer.optimizer.step_count = 0
        pb.lazy_obs = lazy
        smp = pb.sample_inds(inds)
        info, _, _ = trainer.train_policy_on_batch(tuple({"policy_0": x} for x in smp) + (None, None))
        launched = _lib.last_launches()
        assert ("trunk_fwd4_store<16>" in launched) == lazy and ("wgrad_store<4>" in launched) == lazy, launched
        assert ("trunk_fwd4<16>" in launched) != lazy
        out.append((trainer.grad.clone(), trainer.theta.clone(), float(info["loss"])))
    assert np.isfinite(out[0][2]) and out[0][2] == out[1][2]
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_fused_row_chain_on_a_configuration_it_cannot_run_is_an_error():
    """chain_path = 2 together with the wide-state mixer (mixer_path = 3) -- which keeps its stream-K GEMM path -- returns OPE_EINVAL."""
    from offpolicy_amd import _lib
    g = load_golden("qmix_gall_tiny")
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune.update(chain_path=2, mixer_path=3)
    with pytest.raises(_lib.OpeError):
        trainer.train_policy_on_batch(batch_from(buf, g["inds"]))


@pytest.mark.parametrize("chain", [1, 2])
@pytest.mark.parametrize("name", ["qmix_tiny", "qmix_odd"])
def test_forward_intermediates_match_oracle(name, chain):
    """Per-stage check (helps localise a failure): live q values, chosen/target agent q, Q_tot of both mixers."""
    from oracle import qmix_oracle as O
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    trainer.tune["debug"] = 1               # ope_qmix_cfg.debug: keep "q_all" etc. in the workspace
    trainer.tune["chain_path"] = chain
    batch = batch_from(buf, g["inds"])
    trainer.train_policy_on_batch(batch)
    torch.cuda.synchronize()
    orc, _ = oracle_from(g)
    store, _ = reference_store_from(g)
    ob = O.sample_inds(store, g["inds"])
    obs, share, acts, rew, dones, dones_env, avail = [torch.as_tensor(np.ascontiguousarray(x)) for x in ob]
    N, T1, B, D = obs.shape
    T = T1 - 1
    s_obs = torch.cat(list(obs), dim=-2)
    q_all, _ = O.agent_q_forward(orc.agent, s_obs, torch.zeros(N * B, 64))
    got_q = trainer.workspace_view(B, "q_all").view(T1, N * B, dims.act_dim).cpu()
    np.testing.assert_allclose(got_q.numpy(), q_all.numpy(), rtol=1e-4, atol=2e-6)
    got_h = trainer.workspace_view(B, "h").view(T1, N * B, 64).cpu()
    x = O.mlp_trunk(orc.agent, s_obs)
    hseq, _ = O.gru_sequence(orc.agent, x, torch.zeros(N * B, 64))
    np.testing.assert_allclose(got_h.numpy(), hseq.numpy(), rtol=1e-4, atol=2e-6)
    _, (err, keep, q_tot) = orc.q_tot_and_error(orc.agent, orc.mixer, ob)
    got_qtot = trainer.workspace_view(B, "qtot").view(T, B).cpu()
    np.testing.assert_allclose(got_qtot.numpy(), q_tot[..., 0].numpy(), rtol=1e-4, atol=2e-6)
    got_err = trainer.workspace_view(B, "err_abs").view(T, B).cpu()
    np.testing.assert_allclose(got_err.numpy(), err.abs()[..., 0].numpy(), rtol=1e-4, atol=2e-6)


def test_deterministic_bitwise():
    """Same inputs twice -> bit-identical gradient vector (no atomics on the path)."""
    g = load_golden("qmix_tiny")
    outs = []
    for _ in range(2):
        dims, buf, policy, trainer = build_from_fixture(g)
        trainer.train_policy_on_batch(batch_from(buf, g["inds"]))
        outs.append(trainer.grad.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["qmix_tiny", "qmix_shape_layer2", "qmix_var_layer2_d252", "qmix_shape_nofn", "qmix_var_nofn_d252", "qmix_shape_tanh",
                                  "qmix_var_tanh_d252"])
def test_policy_forward_matches_oracle_single_step_and_sequence(name):
    """policy.get_q_values (ope_agent_forward) with a non-zero initial hidden state; with one and with two hidden blocks (layer_N)."""
    from oracle import qmix_oracle as O
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    P = {k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()}
    torch.manual_seed(3)
    obs = torch.randn(5, 7, dims.obs_dim)
    h0 = torch.randn(7, 64) * 0.5
    q_ref, h_ref = O.agent_q_forward(P, obs, h0, use_relu=bool(g["hp_use_relu"]) if "hp_use_relu" in g else True)
    q, h = policy.get_q_values(obs.cuda(), None, h0.cuda())
    np.testing.assert_allclose(q.cpu().numpy(), q_ref.numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref.numpy(), rtol=1e-4, atol=2e-6)
    q1, h1 = policy.get_q_values(obs[0].cuda(), None, h0.cuda())
    np.testing.assert_allclose(q1.cpu().numpy(), q_ref[0].numpy(), rtol=1e-4, atol=2e-6)
    acts, hn, gq = policy.get_actions(obs[0].cuda(), None, h0.cuda(), available_actions=np.ones((7, dims.act_dim)))
    assert acts.shape == (7, dims.act_dim) and np.allclose(acts.sum(-1), 1)
    assert np.array_equal(acts.argmax(-1), q_ref[0].argmax(-1).numpy())


@pytest.mark.parametrize("name", ["qmix_var_d370", "qmix_var_d252"])
def test_single_net_launch_of_the_lds_resident_trunk_matches_oracle(name):
    """A forward over >= 16 384 rows of a width the LDS-resident trunk kernel is built for runs it on ONE net (trunk_fwd4_single: every CU
    holds that net's weights) -- what the recurrent MADDPG / MATD3 actors' live / target / rollout trunks are at BASELINE config 5
    (231 680 rows of MMM2's 370-wide observations; 24 chunks, rows 8-byte aligned) -- here through policy.get_q_values
    (ope_agent_forward), against the oracle's forward on the same rows, with the launch log naming the kernel."""
    from oracle import qmix_oracle as O
    from offpolicy_amd import _lib
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    P = {k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()}
    torch.manual_seed(5)
    rows = 16 * 1024 + 40                  # (a partial last tile)
    obs = torch.randn(2, rows, dims.obs_dim) * 0.7 + 0.1
    h0 = torch.randn(rows, 64) * 0.5
    q_ref, h_ref = O.agent_q_forward(P, obs, h0)
    q, h = policy.get_q_values(obs.cuda(), None, h0.cuda())
    assert "trunk_fwd4_single<%d,0>" % ((dims.obs_dim + 15) // 16) in _lib.last_launches(), _lib.last_launches()
    np.testing.assert_allclose(q.cpu().numpy(), q_ref.numpy(), rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref.numpy(), rtol=1e-4, atol=3e-6)


def test_policy_forward_with_previous_action_input():
    """prev_act_inp (QMixPolicy.py:29-33, 54-58): the rollout forward feeds [obs | previous one-hot action] to the network."""
    from oracle import qmix_oracle as O
    g = load_golden("qmix_tiny_prevact")
    dims, buf, policy, trainer = build_from_fixture(g)
    assert policy.prev_act_inp and policy.q_network_input_dim == dims.obs_dim + dims.act_dim
    P = {k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()}
    torch.manual_seed(4)
    obs = torch.randn(4, 6, dims.obs_dim)
    prev = torch.eye(dims.act_dim)[torch.randint(0, dims.act_dim, (4, 6))]
    h0 = torch.randn(6, 64) * 0.5
    q_ref, h_ref = O.agent_q_forward(P, torch.cat((obs, prev), dim=-1), h0)
    q, h = policy.get_q_values(obs.numpy(), prev.numpy(), h0.cuda())          # numpy inputs, as the runner passes them
    np.testing.assert_allclose(q.cpu().numpy(), q_ref.numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref.numpy(), rtol=1e-4, atol=2e-6)


def _gpu_decisions(trainer, B, dims, avail, saved_only=False):
    """The discrete decisions the HIP step took, read back from its workspace in the oracle's `forced` format (oracle/qmix_oracle.py,
    "discrete decisions"): ReLU masks of the live trunk (one bit per feature, saved for the backward pass), ReLU masks and abs() signs of
    the live mixer's hyper-networks (from the saved post-ReLU / pre-abs activations), the double-Q greedy indices (argmax of the kernel's
    own q values, `q_all`, kept when ope_qmix_cfg.debug is set; first maximum wins on both sides)."""
    T, N, A = dims.episode_length, dims.n_agents, dims.act_dim
    NB = N * B
    bits = torch.arange(64, device="cuda")

    def unpack(name):
        m = trainer.workspace_view(B, name).view(torch.int64)[:(T + 1) * NB]
        return ((m[:, None] >> bits[None, :]) & 1).bool().view(T + 1, NB, 64).cpu()
    d = {"relu1": unpack("mask1"), "relu2_0": unpack("mask2")}
    if not trainer.vdn:
        for key, name in (("hyp_w1", "hw1"), ("hyp_w2", "hw2"), ("hyp_b2", "hb2")):
            d[key] = (trainer.workspace_view(B, name).view(T, B, 64) > 0).cpu()
        if not saved_only:
            d["abs_w1"] = torch.sign(trainer.workspace_view(B, "v1").view(T, B, N * 32)).cpu()
            d["abs_w2"] = torch.sign(trainer.workspace_view(B, "v2").view(T, B, 32)).cpu()
    return d


def _gpu_greedy(trainer, B, dims, avail):
    T, N, A = dims.episode_length, dims.n_agents, dims.act_dim
    q = trainer.workspace_view(B, "q_all").view(T + 1, N * B, A).clone()
    if avail is not None:
        q[avail.reshape(T + 1, N * B, A) == 0] = -1e10
    return q.max(dim=-1)[1].cpu()


@pytest.mark.parametrize("workload,nb,chain", [("3s5z", 32, 2), ("3s5z", 32, 1), ("MMM2", 8, 2), ("3s5z_gall", 32, 0), ("MMM2", 32, 2), ("MMM2", 32, 1)])
def test_full_size_3s5z_matches_oracle_one_step(workload, nb, chain):
    """BASELINE config 4 at full size (N=8, A=14, D=252, S=216, T=150, B=32): one step vs the oracle; the MMM2
    dimensions (N=10, A=18, D=370, S=322, T=180: 8-byte vector paths, the 24-chunk trunk) at B=8 and at B=32; and 3s5z as the
    reference's own launch script runs it (scripts/train_smac_qmix.sh:14-17: --use_global_all_local_state -> S = 216 + 8 * 252 =
    2 232, --gain 1, hard target updates), B=32: the wide-state mixer kernels at their real size.

    EXACT gradient comparison (round 4; replaces the rank-4 SVD escape of round 3). At this size (4.9 M ReLU units, 1.4 M abs() units
    in the mixer, 541 k argmax decisions per step) a few pre-activations lie within float rounding of their decision boundary and the
    CPU and GPU reduction orders may put them on different sides; forward values do not notice, a gradient changes by one rank-one term
    per flipped unit. So the comparison is made at EQUAL decisions: (1) the decisions the HIP step took are read back from its workspace
    (saved ReLU masks, hyper-network activations, q values) and compared with the oracle's own -- at most a handful may differ, and
    they are printed; (2) the oracle's step is re-run with the GPU's decisions forced (oracle `forced=`) and EVERY element of EVERY
    gradient tensor must then agree within 2e-3 of the tensor's max magnitude -- no rank removal, no percentile; (3) the parameters
    after the Adam step agree with the forced oracle's within the bound the gradient difference itself implies."""
    from oracle import qmix_oracle as O
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import DIMS, synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    dims = DIMS[workload]
    args = default_args(gain=1.0, use_soft_update=False) if workload.endswith("_gall") else default_args()
    torch.manual_seed(1)
    np.random.seed(1)
    dev = torch.device("cuda:0")
    policy = QMixPolicy({"args": args, "device": dev}, policy_info_for(dims)["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length)
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, nb, dims.episode_length, True, True, device=dev)
    ep = synth_episodes(np.random.RandomState(0), nb, dims, avail="bernoulli")
    d = as_policy_dicts(ep)
    buf.insert(nb, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    inds = np.arange(nb)
    agent0 = {k: v.detach().cpu().numpy().copy() for k, v in policy.q_network.named_parameters()}
    mixer0 = {k: v.detach().cpu().numpy().copy() for k, v in trainer.mixer.named_parameters()}
    batch = batch_from(buf, inds)
    avail_dev = batch[6]["policy_0"].permute(1, 0, 2, 3).contiguous()         # [T+1, N, B, A]: the kernels' row order (agent * B + b)
    # a first, throw-away step with ope_qmix_cfg.debug (keeps the live q values): only the greedy indices are taken from it
    opt = trainer.optimizer
    snap = (trainer.theta.clone(), trainer.theta_tgt.clone())
    trainer.tune["chain_path"] = chain      # 2: the fused chain, 1: four launches (0: by shape -- the wide-state configuration keeps the separate kernels)
    # every padded row (rounds 1-5's path): the decisions are read back from the workspace in the batch's own row order. The live-row path
    # (what "by shape" picks at 3s5z since round 6) is compared with this one at the end, and with the reference's own numbers in
    # test_gpu_fullsize_reference.py / test_gpu_live_rows.py.
    trainer.tune["live_rows"] = 1
    trainer.tune["debug"] = 1
    trainer.train_policy_on_batch(batch)
    greedy = _gpu_greedy(trainer, nb, dims, avail_dev)
    gpu = _gpu_decisions(trainer, nb, dims, avail_dev)      # (the fused chain keeps the pre-abs mixer weights only under `debug`)
    trainer.tune["debug"] = 0
    trainer.theta.copy_(snap[0]); trainer.theta_tgt.copy_(snap[1]); opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_count = 0
    # the step under test: exactly what bench.py runs
    info, _, _ = trainer.train_policy_on_batch(batch)
    launched = _lib.last_launches()
    if workload == "3s5z":        # the bench line's kernel variants are the ones compared here
        assert "trunk_fwd4<16>" in launched and "gru_fwd4<4>" in launched and "trunk_bwd4" in launched, launched
        assert ("qchain<1,1>" in launched and "mixer_hyp<4>" in launched) if chain == 2 else "mixer_fwd3<14,1>" in launched, launched
    # the decisions this step saved for its own backward pass are those of the throw-away step (same arithmetic, bit for bit)
    again = _gpu_decisions(trainer, nb, dims, avail_dev, saved_only=True)
    for k, v in again.items():
        assert torch.equal(v, gpu[k]), k
    gpu["greedy"] = greedy
    st = {k: (ep[k][:, :, 0] if k == "share_obs" else ep[k]) for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")}
    ob = O.sample_inds(st, inds)
    # (1) the oracle's own step and its own decisions
    own = {}
    orc = O.QMixOracle(agent0, mixer0, dims.n_agents, O.HP())
    out = orc.train_step(ob, fused_gru=True, soft_update=False, record=own)
    np.testing.assert_allclose(float(info["loss"]), out["loss"], rtol=RTOL)
    np.testing.assert_allclose(float(info["grad_norm"]), out["grad_norm"], rtol=RTOL)
    np.testing.assert_allclose(float(info["Q_tot"]), out["Q_tot"], rtol=RTOL, atol=1e-6)
    flips = {k: int((own[k] != gpu[k]).sum()) for k in own}
    units = sum(int(own[k].numel()) for k in own)
    print("decisions that differ between the HIP step and the CPU oracle (of %d): %s" % (units, flips))
    assert sum(flips.values()) <= max(8, units // 200000), flips           # a handful in millions (float rounding at the boundary), not a pattern
    # (2) gradients at equal decisions: every element
    orc = O.QMixOracle(agent0, mixer0, dims.n_agents, O.HP())
    outf = orc.train_step(ob, fused_gru=True, soft_update=False, forced=gpu)
    np.testing.assert_allclose(float(info["loss"]), outf["loss"], rtol=RTOL)
    np.testing.assert_allclose(float(info["grad_norm"]), outf["grad_norm"], rtol=RTOL)
    cnt = float(trainer.grad[trainer.numel + 1])
    got = _flat_named(trainer, trainer.grad[:trainer.numel] / cnt)
    grads = {k: v for k, v in outf["grads"].items() if v is not None}
    worst = {}
    for k, ref in grads.items():
        scale = max(np.abs(ref).max(), 1e-9)
        worst[k] = float(np.abs(got[k] - ref).max() / scale)
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    assert not bad, ("gradient elements beyond 2e-3 of their tensor's max magnitude, at equal decisions", bad)
    # (3) parameters after the first Adam step. (a) The optimizer arithmetic on its own: with zero moments the first step is exactly
    # theta - lr * g / (|g| + eps), g = the clipped gradient -- recomputed here from the GPU's own gradient vector and pre-clip norm,
    # element by element (float rounding only). (b) Against the forced oracle's parameters: f(g) = g / (|g| + eps) has |f'| <= 1 / eps, so
    # two correct implementations differ by at most lr * |g_gpu - g_oracle| / eps per element (+ rounding) -- checked element by element
    # with the ACTUAL clipped-gradient difference of that element.
    lr, eps = args.lr, args.opti_eps
    coef = min(1.0, float(args.max_grad_norm) / (float(info["grad_norm"]) + 1e-6))
    coef_o = min(1.0, float(args.max_grad_norm) / (outf["grad_norm"] + 1e-6))
    live = _flat_named(trainer, trainer.theta)
    init = {("agent/" + k): v for k, v in agent0.items()}
    init.update({("mixer/" + k): v for k, v in mixer0.items()})
    for k, gv in got.items():
        gg = gv.astype(np.float64) * coef
        want = init[k].astype(np.float64) - lr * gg / (np.abs(gg) + eps)
        np.testing.assert_allclose(live[k], want, rtol=0, atol=3e-7, err_msg="Adam step " + k)
    for grp, ref in (("agent/", orc.agent), ("mixer/", orc.mixer)):
        for k, v in ref.items():
            if grp + k not in grads:
                continue                                                     # fc_h: no gradient, untouched on both sides
            dg = np.abs(got[grp + k].astype(np.float64) * coef - grads[grp + k].astype(np.float64) * coef_o)
            dth = np.abs(live[grp + k].astype(np.float64) - v.numpy().astype(np.float64))
            assert (dth <= lr * dg / eps * 1.001 + 3e-7).all(), (grp + k, float((dth - lr * dg / eps).max()))
    # (4) the same step on live rows only (round 6): same per-row arithmetic, another summation order over rows
    if chain != 1 and trainer.vdn is False and workload == "3s5z":
        padded = trainer.grad[:trainer.numel + 4].clone()
        trainer.theta.copy_(snap[0]); trainer.theta_tgt.copy_(snap[1]); opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_count = 0
        trainer.tune["live_rows"] = 2
        info2, _, _ = trainer.train_policy_on_batch(batch)
        assert "qchain_live<1,1>" in _lib.last_launches() and "wgrad2_live<4>" in _lib.last_launches(), _lib.last_launches()
        np.testing.assert_allclose(float(info2["loss"]), float(info["loss"]), rtol=1e-5)
        np.testing.assert_allclose(float(info2["grad_norm"]), float(info["grad_norm"]), rtol=1e-5)
        a_, b_ = padded.cpu().numpy(), trainer.grad[:trainer.numel + 4].cpu().numpy()
        assert a_[trainer.numel + 1] == b_[trainer.numel + 1]
        got2 = _flat_named(trainer, trainer.grad[:trainer.numel] / cnt)
        for k, ref in got.items():
            assert np.abs(got2[k] - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-9), k


def test_runner_call_sequence_with_per():
    """The call sequence of RecRunner.batch_train_q (offpolicy/runner/rnn/base_runner.py:259-284) against our classes:
    sample(beta, p_id) -> train_policy_on_batch -> update_priorities -> soft_target_updates, repeated; loss goes down."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import DIMS, synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import PrioritizedRecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    dims = DIMS["tiny"]
    args = default_args(use_per=True, lr=1e-3)
    torch.manual_seed(2)
    np.random.seed(2)
    dev = torch.device("cuda:0")
    pinfo = policy_info_for(dims)
    policy = QMixPolicy({"args": args, "device": dev}, pinfo["policy_0"])
    trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length)
    buf = PrioritizedRecReplayBuffer(args.per_alpha, pinfo, {"policy_0": [0, 1]}, 24, dims.episode_length, True, True, device=dev)
    d = as_policy_dicts(synth_episodes(np.random.RandomState(0), 24, dims, avail="bernoulli", runner_padding=True))
    buf.insert(24, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    losses = []
    for it in range(30):
        trainer.prep_training()
        sample = buf.sample(8, 0.4, "policy_0")
        info, new_priorities, idxes = trainer.train_policy_on_batch(sample)
        buf.update_priorities(idxes, new_priorities, "policy_0")
        trainer.soft_target_updates()
        losses.append(float(info["loss"]))
        assert np.isfinite(losses[-1]) and new_priorities.shape == (8,) and (new_priorities > 0).all()
    assert np.mean(losses[-5:]) < np.mean(losses[:5])
    assert buf.max_priorities["policy_0"] >= 1.0


def test_time_chunked_two_stream_schedule_matches_single_stream():
    """ope_qmix_cfg.time_chunks > 1 (trainer.tune["time_chunks"]; process default: OPE_CHUNKS) cuts the episode into time chunks and runs the scans on a side stream beside the row-parallel kernels
    (DESIGN.md section 4). Same kernels, same per-row arithmetic: gradients agree with the single-stream schedule to
    summation-order rounding, and are bit-stable from run to run."""
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import DIMS, EnvDims, policy_info_for, synth_episodes
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    d0 = DIMS["3m"]
    dims = EnvDims("3m_t100", d0.n_agents, d0.act_dim, d0.obs_dim, d0.state_dim, 100)
    dev = torch.device("cuda:0")
    pinfo = policy_info_for(dims)
    torch.manual_seed(1)
    policy = QMixPolicy({"args": default_args(), "device": dev}, pinfo["policy_0"])
    trainer = QMix(default_args(), dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=100)
    buf = RecReplayBuffer(pinfo, {"policy_0": list(range(dims.n_agents))}, 16, 100, True, True, device=dev)
    ep = synth_episodes(np.random.RandomState(0), 16, dims, avail="bernoulli")
    buf.insert(16, *[{"policy_0": ep[k]} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")])
    s = buf.policy_buffers["policy_0"].sample_inds(np.arange(8))
    batch = tuple({"policy_0": x} for x in s) + (None, None)
    theta0, tgt0 = trainer.theta.clone(), trainer.theta_tgt.clone()
    grads = {}
    for tag, chunks in (("c1", "1"), ("c3", "3"), ("c3b", "3"), ("c2", "2")):
        trainer.tune["time_chunks"] = int(chunks)
        trainer.theta.copy_(theta0)
        trainer.theta_tgt.copy_(tgt0)
        trainer.optimizer.exp_avg.zero_(); trainer.optimizer.exp_avg_sq.zero_(); trainer.optimizer.step_count = 0
        trainer.train_policy_on_batch(batch)
        torch.cuda.synchronize()
        grads[tag] = trainer.grad.cpu().numpy().copy()
    assert np.array_equal(grads["c3"], grads["c3b"])
    scale = np.abs(grads["c1"]).max()
    for tag in ("c2", "c3"):
        np.testing.assert_allclose(grads[tag], grads["c1"], rtol=0, atol=2e-6 * scale)


@pytest.mark.parametrize("gather_in_graph", [True, False])
def test_graphed_step_matches_reference(gather_in_graph):
    """The captured HIP graph (gather + train_policy_on_batch + soft update, or the training kernels alone behind an eager
    gather) replays the reference's training steps: restore the initial state after the capture warm-up, replay the
    fixture's batch, compare loss / grad_norm / final parameters with the frozen reference outputs."""
    g = load_golden("qmix_tiny")
    dims, buf, policy, trainer = build_from_fixture(g)
    theta0, tgt0 = trainer.theta.clone(), trainer.theta_tgt.clone()
    inds = np.asarray(g["inds"])
    step = trainer.make_graphed_step(buf, len(inds), gather_in_graph=gather_in_graph)
    opt = trainer.optimizer
    # the capture warm-up trains two throw-away steps; make_graphed_step itself must put the state back
    assert torch.equal(trainer.theta, theta0) and torch.equal(trainer.theta_tgt, tgt0)
    assert not opt.exp_avg.any() and not opt.exp_avg_sq.any() and int(opt.step_dev[0].item()) == 0 and opt.step_count == 0
    for s in range(len(g["loss"])):
        info = step(inds)
        np.testing.assert_allclose(float(info["loss"]), g["loss"][s], rtol=RTOL)
        np.testing.assert_allclose(float(info["grad_norm"]), g["grad_norm"][s], rtol=RTOL)
    assert int(opt.step_dev[0].item()) == len(g["loss"]) == opt.step_count
    live, tgt = _flat_named(trainer, trainer.theta), _flat_named(trainer, trainer.theta_tgt)
    for grp, src in (("final_agent/", live), ("final_agent_tgt/", tgt)):
        for k, ref in sub(g, grp).items():
            np.testing.assert_allclose(src["agent/" + k], ref, rtol=0, atol=3e-5, err_msg=grp + k)


def test_vdn_ignores_hypernet_layers_and_tanh_accepts_weight_decay():
    """VDN has no hyper-networks, so `--hypernet_layers 1` must not be refused for it (the reference runs it; same results
    as the default); and `use_ReLU = False` alone leaves no constant slot in the flat vector, so a non-zero args.weight_decay (which QMix's
    Adam ignores anyway, A-8) is no reason to refuse the configuration."""
    from gpu_util import make_args
    from offpolicy_amd.utils.synth import policy_info_for
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    g = load_golden("vdn_tiny")
    dims, buf, policy, trainer = build_from_fixture(g)
    args = make_args(g, hypernet_layers=1)
    dev = torch.device("cuda:0")
    pol1 = QMixPolicy({"args": args, "device": dev}, policy_info_for(dims)["policy_0"])
    tr1 = QMix(args, dims.n_agents, {"policy_0": pol1}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length, vdn=True)
    pol1.q_network.load_state_dict(policy.q_network.state_dict())
    tr1.hard_target_updates()
    batch = batch_from(buf, g["inds"])
    a, _, _ = trainer.train_policy_on_batch(batch)
    b, _, _ = tr1.train_policy_on_batch(batch)
    assert float(a["loss"]) == float(b["loss"]) and torch.equal(trainer.theta, tr1.theta)
    np.testing.assert_allclose(float(a["loss"]), g["loss"][0], rtol=RTOL)
    gt = load_golden("qmix_shape_tanh")
    args_t = make_args(gt, weight_decay=0.01)
    pol_t = QMixPolicy({"args": args_t, "device": dev}, policy_info_for(fixture_dims_of(gt))["policy_0"])
    QMix(args_t, int(gt["dims"][0]), {"policy_0": pol_t}, lambda a: "policy_0", device=dev, episode_length=int(gt["dims"][4]))
    with pytest.raises(NotImplementedError):
        args_n = make_args(load_golden("qmix_shape_nofn"), weight_decay=0.01)
        pol_n = QMixPolicy({"args": args_n, "device": dev}, policy_info_for(fixture_dims_of(gt))["policy_0"])
        QMix(args_n, int(gt["dims"][0]), {"policy_0": pol_n}, lambda a: "policy_0", device=dev, episode_length=int(gt["dims"][4]))


def fixture_dims_of(g):
    from golden_util import fixture_dims
    return fixture_dims(g)


def test_multi_discrete_policy_forward_and_actions_match_oracle():
    """MultiDiscrete (round 5): the rollout forward returns the list of per-head q tensors (upstream's ACTLayer, act.py:23-31) -- the column
    blocks of the one stacked head the kernels evaluate --, q_values_from_actions picks one q per head from the concatenated one-hot
    blocks, and greedy actions are chosen per head."""
    from oracle import qmix_oracle as O
    g = load_golden("qmix_md_odd_huber_per")
    dims, buf, policy, trainer = build_from_fixture(g)
    heads = [int(x) for x in g["multi_discrete"]]
    P = {k: torch.as_tensor(v) for k, v in sub(g, "agent/").items()}
    torch.manual_seed(3)
    obs = torch.randn(5, 7, dims.obs_dim)
    h0 = torch.randn(7, 64) * 0.5
    q_ref, h_ref = O.agent_q_forward(P, obs, h0)                      # [5, 7, sum(heads)]: the heads' blocks side by side
    q, h = policy.get_q_values(obs.cuda(), None, h0.cuda())
    assert isinstance(q, list) and [int(x.shape[-1]) for x in q] == heads
    np.testing.assert_allclose(torch.cat(q, dim=-1).cpu().numpy(), q_ref.numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), h_ref.numpy(), rtol=1e-4, atol=2e-6)
    acts, hn, gq = policy.get_actions(obs[0].cuda(), None, h0.cuda())
    assert acts.shape == (7, sum(heads)) and tuple(gq.shape) == (7, len(heads))
    lo = 0
    for k, d in enumerate(heads):
        assert np.array_equal(acts[:, lo:lo + d].argmax(-1), q_ref[0][:, lo:lo + d].argmax(-1).numpy()) and np.allclose(acts[:, lo:lo + d].sum(-1), 1)
        np.testing.assert_allclose(gq[:, k].cpu().numpy(), q_ref[0][:, lo:lo + d].max(-1)[0].numpy(), rtol=1e-4, atol=2e-6)
        lo += d
    taken, _ = policy.get_q_values(obs.cuda(), None, h0.cuda(), action_batch=torch.as_tensor(np.tile(acts[None], (5, 1, 1))))
    assert tuple(taken.shape) == (5, 7, len(heads))


@pytest.mark.parametrize("name", ["qmix_3m_katA", "qmix_odd"])
def test_long_horizon_tracks_the_oracle(name):
    """150 consecutive updates (fresh index draws, Adam's bias correction running on, Polyak steps in between) against the oracle stepping on
    the same indices: per-step loss / grad_norm / Q_tot stay together (fp32 round-off compounds, so the band widens with the step count), and
    so do the parameters at the end. What three-step fixtures cannot show: state carried from step to step (optimizer moments, step counter,
    target networks, cached plans / workspaces)."""
    from oracle import qmix_oracle as O
    from golden_util import record_errors
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    orc, _ = oracle_from(g)
    store, _ = reference_store_from(g)
    n_ep, B = len(g["idx_range"]), len(g["inds"])
    rng = np.random.RandomState(5)
    worst = 0.0
    threads = torch.get_num_threads()
    torch.set_num_threads(8)      # (the oracle's small CPU tensors: a GPU box's full core count costs 1.5 s per step in thread hand-offs)
    try:
        for st in range(150):
            inds = rng.choice(n_ep, B)
            info, _, _ = trainer.train_policy_on_batch(batch_from(buf, inds))
            trainer.soft_target_updates()
            out = orc.train_step(O.sample_inds(store, inds))
            tol = 5e-5 * (1 + st / 10.0)      # (measured: 1.7e-6 / 6.3e-6 in these units, profiles/r05_parity_errors.txt)
            for k_e, k_o in (("loss", "loss"), ("grad_norm", "grad_norm"), ("Q_tot", "Q_tot")):
                a, b = float(info[k_e]), float(out[k_o])
                worst = max(worst, abs(a - b) / max(abs(b), 1e-3) / (1 + st / 10.0))
                assert abs(a - b) <= tol * max(abs(b), 1e-3), (st, k_e, a, b)
    finally:
        torch.set_num_threads(threads)
    for k, v in policy.q_network.named_parameters():
        np.testing.assert_allclose(v.detach().cpu().numpy(), orc.agent[k].numpy(), rtol=0, atol=2e-4, err_msg=k)
    record_errors("qmix_long_horizon:" + name, {"worst_scaled_rel": worst})


def test_weight_gradient_table_the_register_blocked_kernel_cannot_plan_falls_back_before_the_first_launch():
    """one-layer hyper-networks with 16 agents and a wide state make hyper_w1 a 512 x 800 problem = 28 units of the register-blocked
    weight-gradient launch, 46 with the others -- more than its table holds. "By shape" (wgrad_path 0) must then run the whole step on the
    tile-per-wave kernel (same results as pinning it), decided BEFORE the first launch; pinning wgrad2 (wgrad_path 2) is an error with
    nothing launched."""
    from offpolicy_amd import _lib
    from offpolicy_amd.config import default_args
    from offpolicy_amd.utils.synth import EnvDims, synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.algorithms.qmix.algorithm.QMixPolicy import QMixPolicy
    from offpolicy_amd.algorithms.qmix.qmix import QMix
    dims, B = EnvDims("wide16", 16, 6, 16, 800, 5), 4
    args = default_args(hypernet_layers=1)
    out = {}
    for path in (0, 1):
        torch.manual_seed(5)
        np.random.seed(5)
        dev = torch.device("cuda:0")
        policy = QMixPolicy({"args": args, "device": dev}, policy_info_for(dims)["policy_0"])
        trainer = QMix(args, dims.n_agents, {"policy_0": policy}, lambda a: "policy_0", device=dev, episode_length=dims.episode_length)
        buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, B, dims.episode_length, True, True, device=dev)
        d = as_policy_dicts(synth_episodes(np.random.RandomState(0), B, dims, avail="bernoulli"))
        buf.insert(B, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
        batch = batch_from(buf, np.arange(B))
        trainer.tune["wgrad_path"] = path
        info, _, _ = trainer.train_policy_on_batch(batch)
        launched = _lib.last_launches()
        assert any(k.startswith("wgrad<") for k in launched) and not any(k.startswith("wgrad2") for k in launched), launched
        assert np.isfinite(float(info["loss"]))
        out[path] = trainer.grad.clone()
        if path == 0:
            trainer.tune["wgrad_path"] = 2
            with pytest.raises(_lib.OpeError):
                trainer.train_policy_on_batch(batch)
            assert _lib.last_launches() == []
    assert torch.equal(out[0], out[1])
