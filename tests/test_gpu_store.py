"""-m gpu: replay store insert/gather through the C-ABI, bit-exact against the oracle / reference fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from golden_util import reference_store_from
from gpu_util import build_from_fixture

pytestmark = pytest.mark.gpu
FIELDS = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")


@pytest.mark.parametrize("name", ["qmix_tiny", "qmix_tiny_huber_per", "qmix_odd", "vdn_tiny", "qmix_tiny_pershare"])
def test_gather_bit_exact_vs_reference_fixture(name):
    """insert (with ring wrap) + sample_inds == what the reference's RecPolicyBuffer returned, bit for bit."""
    g = load_golden(name)
    dims, buf, _, _ = build_from_fixture(g)
    got = buf.policy_buffers["policy_0"].sample_inds(g["inds"])
    for k, a in zip(FIELDS, got):
        ref = g["batch/" + k]
        assert tuple(a.shape) == ref.shape, k
        assert np.array_equal(a.cpu().numpy(), ref), k


@pytest.mark.parametrize("dims_name,B", [("3m", 32), ("3s5z", 32), ("MMM2", 8), ("simple_spread", 16)])
def test_gather_matches_oracle_at_scale(dims_name, B):
    """Full-size configs: HIP gather vs the numpy oracle on the same seeded store (bit-exact), repeated + wrapped indices."""
    from oracle.qmix_oracle import sample_inds
    from offpolicy_amd.utils.synth import DIMS, synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    dims = DIMS[dims_name]
    cap = 40
    rng = np.random.RandomState(5)
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, True, True,
                          device="cuda:0")
    T, N, A, D, S = dims.episode_length, dims.n_agents, dims.act_dim, dims.obs_dim, dims.state_dim
    st = dict(obs=np.zeros((T + 1, cap, N, D), np.float32), share_obs=np.zeros((T + 1, cap, S), np.float32),
              acts=np.zeros((T, cap, N, A), np.float32), avail_acts=np.ones((T + 1, cap, N, A), np.float32),
              rewards=np.zeros((T, cap, N, 1), np.float32), dones=np.ones((T, cap, N, 1), np.float32),
              dones_env=np.ones((T, cap, 1), np.float32))
    for n_ins in (25, 30):            # second insert wraps around
        ep = synth_episodes(rng, n_ins, dims, avail="bernoulli")
        d = as_policy_dicts(ep)
        idx = buf.insert(n_ins, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
        for k in FIELDS:
            st[k][:, idx] = ep[k][:, :, 0] if k == "share_obs" else ep[k]
    inds = rng.randint(0, len(buf), size=B)
    inds[1] = inds[0]
    got = buf.policy_buffers["policy_0"].sample_inds(inds)
    want = sample_inds(st, inds)
    for k, a, w in zip(FIELDS, got, want):
        assert np.array_equal(a.cpu().numpy(), w), k


def test_gather_untouched_slots_keep_reference_defaults():
    """Slots never written read back as the reference's initial values (zeros; ones for avail/dones/dones_env)."""
    from offpolicy_amd.utils.synth import DIMS, policy_info_for
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    dims = DIMS["tiny"]
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": [0, 1]}, 4, dims.episode_length, True, True, device="cuda:0")
    obs, share, acts, rew, dones, dones_env, avail = buf.policy_buffers["policy_0"].sample_inds(np.array([0, 3]))
    assert float(obs.abs().sum()) == 0 and float(acts.abs().sum()) == 0 and float(rew.abs().sum()) == 0
    assert bool((avail == 1).all()) and bool((dones == 1).all()) and bool((dones_env == 1).all())


def test_prioritized_buffer_flow():
    """PrioritizedRecReplayBuffer: every inserted slot gets max_priority**alpha (SURVEY A-3 fix), proportional sampling
    and importance weights follow rec_buffer.py:278-304, update_priorities follows 306-324 incl. its assertions."""
    from offpolicy_amd.utils.synth import DIMS, synth_episodes, policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import PrioritizedRecReplayBuffer
    dims = DIMS["tiny"]
    alpha, beta = 0.6, 0.4
    buf = PrioritizedRecReplayBuffer(alpha, policy_info_for(dims), {"policy_0": [0, 1]}, 6, dims.episode_length, True, True,
                                     device="cuda:0")
    rng = np.random.RandomState(0)
    for n in (1, 4, 3):               # a single-episode insert (crashes upstream), then a wrap-around
        d = as_policy_dicts(synth_episodes(rng, n, dims))
        idx = buf.insert(n, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
        assert np.allclose(buf._it_sums["policy_0"][idx], 1.0)
    assert len(buf) == 6
    np.testing.assert_allclose(buf._it_sums["policy_0"].sum(), 6.0)
    np.random.seed(4)
    out = buf.sample(4, beta=beta, p_id="policy_0")
    weights, inds = out[7], out[8]
    assert out[0]["policy_0"].shape == (dims.n_agents, dims.episode_length + 1, 4, dims.obs_dim)
    np.testing.assert_allclose(weights, np.ones(4))           # equal priorities -> all weights 1
    prios = np.array([0.5, 2.0, 1.0, 3.0])
    buf.update_priorities(inds, prios, p_id="policy_0")
    assert buf.max_priorities["policy_0"] == 3.0
    want = np.ones(6)
    for i, p in zip(inds, prios):
        want[i] = p ** alpha
    np.testing.assert_allclose(buf._it_sums["policy_0"][np.arange(6)], want)
    np.testing.assert_allclose(buf._it_mins["policy_0"].min(), want.min())
    np.random.seed(5)
    w2, i2 = buf.sample(5, beta=beta, p_id="policy_0")[7:]
    p_s = want[i2] / want.sum()
    np.testing.assert_allclose(w2, (p_s * 6) ** (-beta) / ((want.min() / want.sum() * 6) ** (-beta)))
    with pytest.raises(AssertionError):
        buf.update_priorities(inds, np.array([1.0, 0.0, 1.0, 1.0]), p_id="policy_0")     # priorities must be > 0
    with pytest.raises(AssertionError):
        buf.sample(6, beta=beta, p_id="policy_0")                                        # needs len > batch_size
    with pytest.raises(AssertionError):
        buf.sample(2, beta=0, p_id="policy_0")
    d = as_policy_dicts(synth_episodes(rng, 2, dims))
    new_idx = buf.insert(2, d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    np.testing.assert_allclose(buf._it_sums["policy_0"][new_idx], 3.0 ** alpha)          # new episodes get the running max


def test_reward_normalisation_matches_reference_buffers():
    """use_reward_normalization=True: device statistics + in-place normalisation vs the real reference buffers' output
    (partially filled ring, wrapped ring, transitions)."""
    from conftest import load_golden
    from offpolicy_amd.utils.synth import EnvDims, policy_info_for
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.utils.mlp_buffer import MlpReplayBuffer
    from oracle import reward_norm as RN
    g = load_golden("reward_norm")
    n, a, d, s, T = [int(x) for x in g["rec_dims"]]
    dims = EnvDims("rn", n, a, d, s, T)
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(n))}, 8, T, True, True, True, device="cuda:0")
    keys = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")
    for tag in ("a", "b"):
        r = buf.insert(len(g["rec_%s/idx_range" % tag]), *[{"policy_0": g["rec_%s/ep/%s" % (tag, k)]} for k in keys])
        assert np.array_equal(r, g["rec_%s/idx_range" % tag])
        pb = buf.policy_buffers["policy_0"]
        got = pb.sample_inds(g["rec_%s/inds" % tag])[3].cpu().numpy()
        np.testing.assert_allclose(got, g["rec_%s/rewards" % tag], rtol=2e-5, atol=2e-6)
        # statistics against the oracle on the reference's time-major view of our store
        mean, std = RN.episode_reward_stats(pb.rewards.cpu().numpy().transpose(1, 0, 2, 3), pb.dones_env.cpu().numpy().transpose(1, 0, 2),
                                            pb.filled_i)
        st = pb.reward_stats().cpu().numpy()
        np.testing.assert_allclose(st[:2], [mean, std], rtol=1e-6)
    n, a, d, s, _ = [int(x) for x in g["mlp_dims"]]
    tdims = EnvDims("rnt", n, a, d, s, 1)
    from test_mlp_oracle_golden import T_KEYS
    mbuf = MlpReplayBuffer(policy_info_for(tdims), {"policy_0": list(range(n))}, 16, True, True, True, device="cuda:0")
    mbuf.insert(len(g["mlp/idx_range"]), *[{"policy_0": g["mlp/tr/" + k]} for k in T_KEYS])
    got = mbuf.policy_buffers["policy_0"].sample_inds(g["mlp/inds"])[3].cpu().numpy()
    np.testing.assert_allclose(got, g["mlp/rewards"], rtol=2e-5, atol=2e-6)


def test_device_per_trees_match_host_trees():
    """Device sum/min trees (ope_per_tree_*) against the host segment trees through a random sequence of inserts, priority
    updates with duplicate indices, and proportional samples driven by the same uniform draws."""
    from offpolicy_amd.utils.device_per import DevicePerTree
    from offpolicy_amd.utils.segment_tree import SumSegmentTree, MinSegmentTree
    cap, alpha = 64, 0.6
    dt = DevicePerTree(cap, alpha, "cuda:0")
    hs, hm = SumSegmentTree(cap), MinSegmentTree(cap)
    rng = np.random.RandomState(4)
    max_prio = 1.0
    filled = 0
    for it in range(12):
        n = int(rng.randint(1, 9))
        idx = (np.arange(n) + filled) % cap                 # ring insert
        filled = min(cap, filled + n)
        dt.set_to_max(idx)
        hs[idx] = max_prio ** alpha
        hm[idx] = max_prio ** alpha
        upd = rng.randint(0, filled, size=int(rng.randint(2, 12)))   # duplicates on purpose
        pr = rng.uniform(0.05, 3.0, size=len(upd)).astype(np.float32)
        dt.set(upd, pr)
        hs[upd] = pr.astype(np.float64) ** alpha
        hm[upd] = pr.astype(np.float64) ** alpha
        max_prio = max(max_prio, float(pr.max()))
        sl, ml, mp = dt.leaves()
        np.testing.assert_allclose(sl, hs[np.arange(cap)], rtol=1e-12)
        np.testing.assert_allclose(ml[:filled], hm[np.arange(filled)], rtol=1e-12)
        np.testing.assert_allclose(mp, max_prio, rtol=1e-7)
        rs, rm = dt.roots()
        np.testing.assert_allclose([rs, rm], [hs.sum(), hm.min()], rtol=1e-12)
        if filled > 4:
            B, beta = 7, 0.4 + 0.05 * it
            mass01 = rng.random_sample(B)
            inds, w = dt.sample(mass01, filled, beta)
            ref = hs.find_prefixsum_idx(mass01 * hs.sum(0, filled - 1))
            assert np.array_equal(inds.cpu().numpy(), ref)
            p_min = hm.min() / hs.sum()
            ref_w = (hs[ref] / hs.sum() * filled) ** (-beta) / (p_min * filled) ** (-beta)
            np.testing.assert_allclose(w.cpu().numpy(), ref_w, rtol=2e-6)


def test_prioritized_buffer_with_device_trees_trains_without_host_round_trip():
    """PrioritizedRecReplayBuffer(device_tree=True): sample -> QMix.train_policy_on_batch -> update_priorities with every PER
    quantity a device tensor; same indices / weights / priorities as the host-tree buffer fed the same random draws."""
    from conftest import load_golden
    from gpu_util import build_from_fixture, make_args
    from golden_util import fixture_dims, fixture_episodes, EP_KEYS
    from offpolicy_amd.utils.synth import policy_info_for, as_policy_dicts
    from offpolicy_amd.utils.rec_buffer import PrioritizedRecReplayBuffer
    g = load_golden("qmix_tiny_huber_per")
    dims, _, policy, trainer = build_from_fixture(g)
    pinfo = policy_info_for(dims)
    ep = as_policy_dicts(fixture_episodes(g))
    bufs = {}
    for mode in (False, True):
        b = PrioritizedRecReplayBuffer(0.6, pinfo, {"policy_0": list(range(dims.n_agents))}, len(g["idx_range"]), dims.episode_length, True, True,
                                       device="cuda:0", device_tree=mode)
        b.insert(len(g["idx_range"]), *[ep[k] for k in EP_KEYS])
        bufs[mode] = b
    theta0, tgt0 = trainer.theta.clone(), trainer.theta_tgt.clone()
    out = {}
    for mode in (False, True):
        trainer.theta.copy_(theta0); trainer.theta_tgt.copy_(tgt0)
        trainer.optimizer.exp_avg.zero_(); trainer.optimizer.exp_avg_sq.zero_(); trainer.optimizer.step_count = 0
        np.random.seed(11)
        rec = []
        for step in range(3):
            batch = bufs[mode].sample(3, beta=0.5, p_id="policy_0")
            info, prio, idxes = trainer.train_policy_on_batch(batch)
            if mode:
                assert torch.is_tensor(prio) and prio.is_cuda and torch.is_tensor(idxes) and idxes.is_cuda and torch.is_tensor(batch[7])
            bufs[mode].update_priorities(idxes, prio, "policy_0")
            to_np = lambda x: x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
            rec.append((to_np(idxes), to_np(batch[7]), to_np(prio), float(info["loss"])))
        out[mode] = rec
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a[0], b[0])
        np.testing.assert_allclose(a[1], b[1], rtol=2e-6)
        np.testing.assert_allclose(a[2], b[2], rtol=2e-6)
        np.testing.assert_allclose(a[3], b[3], rtol=1e-6)


@pytest.mark.parametrize("dims_name,B", [("MMM2", 128), ("3m", 7), ("3s5z", 33), ("simple_spread", 256)])
def test_gather_kernel_variants_and_tile_path_are_bit_identical(dims_name, B):
    """Every A/B variant of the gather (row path for short rows vs the LDS-transposing tile path, non-temporal loads /
    stores, 4 / 8 / 16 rows in flight, block sizes, XCD run lengths) returns the same bytes as a torch index_select of
    the store -- at config 5's batch (B=128: tiles with E=128), an odd batch, and dims whose runs are not 16-byte aligned."""
    from offpolicy_amd import _lib
    from offpolicy_amd.utils.synth import DIMS, policy_info_for
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    dims = DIMS[dims_name]
    cap = 48 if dims_name != "simple_spread" else 512
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, True, True, device="cuda:0")
    pb = buf.policy_buffers["policy_0"]
    pb._ring.filled_i = cap
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    for k in FIELDS:
        getattr(pb, k).normal_(generator=gen)
    inds = np.random.RandomState(3).randint(0, cap, size=B)
    di = torch.as_tensor(inds, device="cuda:0")
    want = [getattr(pb, k).index_select(0, di) for k in FIELDS]      # [B, T(+1), (N,) dim]
    want = [w.permute(2, 1, 0, 3) if w.dim() == 4 else w.permute(1, 0, 2) for w in want]   # -> [N, T(+1), B, dim] / [T(+1), B, dim]
    # the knobs travel PER STORE with the call (ope_gather_tune / ope_store_gather_tuned), not through process state
    for floats, xcd, unroll, nt, small, tile in [(2048, 8, 8, 0, 1, 4096), (2048, 8, 8, 0, 0, 4096), (512, 1, 4, 0, 1, 1024), (8192, 4, 16, 0, 1, 7168),
                                                 (2048, 8, 8, 1, 1, 2048), (4096, 8, 8, 2, 1, 512), (24576, 16, 8, 3, 0, 4096)]:
        pb.gather_tune = dict(floats_per_block=floats, xcd_run=xcd, unroll=unroll, nontemporal=1 + nt, small_tiles=1 if small else 2, tile_floats=tile)
        for src in (inds, di):        # host indices (kernel-argument block) and device-resident indices
            got = pb.sample_inds(src)
            for k, a, w in zip(FIELDS, got, want):
                assert tuple(a.shape) == tuple(w.shape), (k, a.shape, w.shape)
                assert torch.equal(a, w), (k, floats, xcd, unroll, nt, small, tile, torch.is_tensor(src))
    pb.gather_tune = None
    got = pb.sample_inds(inds)            # and the defaults again
    for k, a, w in zip(FIELDS, got, want):
        assert torch.equal(a, w), k


def test_out_of_range_indices_are_skipped_and_flagged():
    """Host index arrays are range-checked like numpy fancy indexing (IndexError, negatives wrap); DEVICE index tensors are
    checked in the kernel: the offending rows are not read (no out-of-bounds access), the rest of the batch is right, and
    check_indices() raises afterwards."""
    from offpolicy_amd.utils.synth import DIMS, policy_info_for
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    dims = DIMS["tiny"]
    cap = 6
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, True, True, device="cuda:0")
    pb = buf.policy_buffers["policy_0"]
    pb._ring.filled_i = cap
    for k in FIELDS:
        getattr(pb, k).normal_()
    with pytest.raises(IndexError):
        pb.sample_inds(np.array([0, cap]))
    a = pb.sample_inds(np.array([-1, 2]))
    b = pb.sample_inds(np.array([cap - 1, 2]))
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    pb.check_indices()                                      # nothing flagged so far
    good = pb.sample_inds(np.array([1, 3, 5, 0]))
    out = pb.alloc_batch(4)
    for v in out.values():
        v.fill_(-7.0)
    got = pb.sample_inds(torch.tensor([1, 3, 10 ** 9, 0], dtype=torch.int64, device="cuda:0"), out=out)
    torch.cuda.synchronize()
    for k, g, w in zip(FIELDS, got, good):
        bdim = g.dim() - 2
        keep = [0, 1, 3]
        assert torch.equal(g.index_select(bdim, torch.tensor(keep, device=g.device)), w.index_select(bdim, torch.tensor(keep, device=g.device))), k
    with pytest.raises(IndexError):
        pb.check_indices()
    pb.check_indices()                                      # flag cleared by the raise


def test_device_sampled_gather_is_uniform_in_range_and_matches_index_gather():
    """ope_store_gather_sampled (RecPolicyBuffer.sample_device): sample(batch) with the uniform draw inside the gather kernel.
    The drawn indices lie in [0, filled) -- NOT the capacity: the buffer is half full --, the gathered batch is bit-identical
    to sample_inds(those indices), a different counter value gives different indices, the same one the same, and over 64 draws
    of 512 the histogram over the 40 filled slots passes a chi-square test (reference: np.random.choice(filled, batch),
    rec_buffer.py:86)."""
    import torch
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_episodes
    dims = DIMS["3m"]
    cap, filled, B = 80, 40, 512
    buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(dims.n_agents))}, cap, dims.episode_length, True, True, device="cuda:0")
    ep = synth_episodes(np.random.RandomState(3), filled, dims)
    buf.insert(filled, *[{"policy_0": ep[k]} for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")])
    pb = buf.policy_buffers["policy_0"]
    ctr = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    got, inds = pb.sample_device(B, seed=12345, counter=ctr)
    torch.cuda.synchronize()
    iv = inds.cpu().numpy()
    assert iv.min() >= 0 and iv.max() < filled
    ref = pb.sample_inds(iv)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    _, again = pb.sample_device(B, seed=12345, counter=ctr)
    assert torch.equal(again, inds)
    counts = np.zeros(filled)
    for t in range(64):
        ctr[0] = t + 1
        _, ii = pb.sample_device(B, seed=12345, counter=ctr)
        v = ii.cpu().numpy()
        if t == 0:
            assert not np.array_equal(v, iv)
        counts += np.bincount(v, minlength=filled)
    expect = 64 * B / filled
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 39 + 5 * np.sqrt(2 * 39), chi2          # 39 degrees of freedom: mean 39, sd 8.8


@pytest.mark.parametrize("name,index_source", [("qmix_tiny", "host"), ("qmix_odd", "device"), ("qmix_tiny_huber_per", "sampled")])
def test_observations_left_in_the_store_are_the_gathered_ones(name, index_source):
    """RecPolicyBuffer.lazy_obs (round 4; SURVEY.md 8(d) "fused into the first consumer"): sample_inds copies every field but obs and
    returns a StoreObs in its place. Everything a caller can do with the reference's array (rec_buffer.py:206-238) still gives the
    reference's values: the other six fields are bit-identical to the fixture, the StoreObs materialises to the fixture's obs, numpy /
    indexing / .cpu() go through the same tensor, the index list the row readers use is the one that was sampled -- and a batch kept
    across an insert() refuses to be read instead of silently changing."""
    from offpolicy_amd.utils.rec_buffer import StoreObs, StaleBatchError
    g = load_golden(name)
    dims, buf, _, _ = build_from_fixture(g)
    pb = buf.policy_buffers["policy_0"]
    inds = np.asarray(g["inds"], dtype=np.int64)
    pb.lazy_obs = True
    if index_source == "host":
        got = pb.sample_inds(inds)
    elif index_source == "device":
        got = pb.sample_inds(torch.as_tensor(inds).cuda())
    else:
        counter = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        got, drawn = pb.sample_device(len(inds), 77, counter)
        pb.lazy_obs = False
        want, drawn2 = pb.sample_device(len(inds), 77, counter)
        pb.lazy_obs = True
        assert torch.equal(drawn, drawn2) and isinstance(got[0], StoreObs)
        assert torch.equal(got[0].materialize(), want[0])
        for a, w in zip(got[1:], want[1:]):
            assert (a is None and w is None) or torch.equal(a, w)
        return
    obs = got[0]
    assert isinstance(obs, StoreObs) and tuple(obs.shape) == g["batch/obs"].shape
    assert np.array_equal(obs.inds.cpu().numpy(), inds)
    for k, a in zip(FIELDS[1:], got[1:]):
        assert np.array_equal(a.cpu().numpy(), g["batch/" + k]), k
    assert np.array_equal(np.asarray(obs), g["batch/obs"])
    assert np.array_equal(obs.cpu().numpy(), g["batch/obs"]) and np.array_equal(obs[1].cpu().numpy(), g["batch/obs"][1])
    pb.check_indices()
    # a later ring write invalidates the rows the object points at
    fresh = pb.sample_inds(inds)[0]
    from golden_util import fixture_episodes
    from offpolicy_amd.utils.synth import as_policy_dicts
    d = as_policy_dicts(fixture_episodes(g))
    buf.insert(len(g["idx_range"]), d["obs"], d["share_obs"], d["acts"], d["rewards"], d["dones"], d["dones_env"], d["avail_acts"])
    with pytest.raises(StaleBatchError):
        fresh.materialize()
    with pytest.raises(StaleBatchError):
        fresh.ref()
    assert np.array_equal(obs.cpu().numpy(), g["batch/obs"])       # (materialised before the insert: a plain tensor now)


def test_device_synthesised_store_has_the_shape_of_host_synthesised_episodes():
    """bench.py fills its 5 000-episode store on the device (utils/synth.synth_fill_device) so that eight ranks do not each push 7.5 GB of
    host-generated numbers: the result must look like `synth_episodes` + insert -- one-hot actions, rewards shared by the agents, dones_env
    a step function that reaches 1 between T/2 - 1 and T - 1 and stays there, dones = dones_env per agent, action 0 always available, ring
    counters advanced with wrap-around -- and be the same on every call with the same seed."""
    from offpolicy_amd.utils.synth import DIMS, policy_info_for, synth_fill_device
    from offpolicy_amd.utils.rec_buffer import RecReplayBuffer
    dims = DIMS["3m"]
    T, N, A = dims.episode_length, dims.n_agents, dims.act_dim
    outs = []
    for _ in range(2):
        buf = RecReplayBuffer(policy_info_for(dims), {"policy_0": list(range(N))}, 48, T, True, True, device="cuda:0")
        pb = buf.policy_buffers["policy_0"]
        synth_fill_device(pb, 60, dims, seed=100, chunk=25)          # 60 episodes into 48 slots: wraps
        assert len(buf) == 48 and pb.filled_i == 48 and pb.current_i == 12
        outs.append([x.clone() for x in (pb.obs, pb.share_obs, pb.acts, pb.rewards, pb.dones, pb.dones_env, pb.avail_acts)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    obs, share, acts, rew, dones, de, av = outs[0]
    assert abs(float(obs.mean())) < 0.01 and abs(float(obs.std()) - 1.0) < 0.01
    assert torch.equal(acts.sum(-1), torch.ones_like(acts[..., 0])) and set(acts.unique().tolist()) == {0.0, 1.0}
    assert torch.equal(rew, rew[:, :, :1].expand_as(rew))
    d = de[:, :, 0]
    assert torch.all(d[:, 1:] >= d[:, :-1]) and torch.all(d[:, -1] == 1)
    L = (d == 0).sum(1) + 1
    assert int(L.min()) >= T // 2 and int(L.max()) <= T
    assert torch.equal(dones, de[:, :, None, :].expand_as(dones))
    assert torch.all(av[..., 0] == 1) and 0.75 < float(av[..., 1:].mean()) < 0.85
    s = pb.sample_inds(np.array([0, 47, 12]))
    assert tuple(s[0].shape) == (N, T + 1, 3, dims.obs_dim)
