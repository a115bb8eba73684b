"""-m gpu: the data-parallel path with the HIP trainers. Two ranks share cuda:0 (the GPU box has one GPU) over gloo; each
back-propagates its half of the sampled episodes through the C-ABI, the flat [grads | loss_sum | mask_count | qtot_sum]
vector goes through ONE all-reduce (the one-shot xGMI kernel through HIP IPC when it verifies on this box, else the process
group's), and the post-step parameters must equal the single-process full-batch step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_shot_allreduce_world1_roundtrip_is_exact():
    """world == 1: the vector goes out to the own slot and back (both parities, ragged tail, max size): bit-exact."""
    from offpolicy_amd.dist import OneShotAllreduce
    dev = torch.device("cuda:0")
    ar = OneShotAllreduce(dev, 0, 1)
    g = torch.Generator(device="cpu").manual_seed(0)
    for n in (1, 5, 1023, 1024, 118795, 1 << 18, 7, 118795):
        x = torch.randn(n, generator=g).to(dev)
        want = x.clone()
        ar(x)
        assert torch.equal(x, want), n
    assert not ar.timed_out() and ar.epoch == 8
    # the host-epoch entry point (launch argument: eager callers that keep their own counter) on a second context
    import ctypes as C
    from offpolicy_amd import _lib
    ar2 = OneShotAllreduce(dev, 0, 1)
    for e, n in ((1, 777), (2, 4096), (3, 5)):
        x = torch.randn(n, generator=g).to(dev)
        want = x.clone()
        _lib.check(_lib.lib.ope_allreduce_flat(C.byref(ar2.ctx), e, _lib.ptr(x), n, _lib.ptr(ar2.status), _lib.current_stream()), "ope_allreduce_flat")
        assert torch.equal(x, want), (e, n)
    assert _lib.lib.ope_allreduce_flat(C.byref(ar2.ctx), 0, _lib.ptr(x), 5, _lib.ptr(ar2.status), _lib.current_stream()) != 0      # epoch 0 is refused
    assert not ar2.timed_out() and ar2.epoch == 0
    ar2.close()
    ar.close()


def _qmix_worker(rank, world, port, name, mode, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OPE_ALLREDUCE=mode)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden
        from gpu_util import build_from_fixture, batch_from
        from offpolicy_amd import dist as opdist
        fast = opdist.setup_fast_allreduce(torch.device("cuda:0"))
        g = load_golden(name)
        dims, buf, policy, trainer = build_from_fixture(g)
        inds = np.asarray(g["inds"])[:4]
        infos = []
        for _ in range(2):
            info, _, _ = trainer.train_policy_on_batch(batch_from(buf, opdist.shard_indices(inds)))
            trainer.soft_target_updates()
            infos.append([float(info[k]) for k in ("loss", "grad_norm", "Q_tot")])
        torch.cuda.synchronize()
        bad = opdist._fast.timed_out() if fast else False
        out_q.put((rank, bool(fast), bad, trainer.theta.cpu().numpy(), trainer.theta_tgt.cpu().numpy(), np.asarray(infos)))
    finally:
        torch.distributed.destroy_process_group()


def _rddpg_worker(rank, world, port, name, mode, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OPE_ALLREDUCE=mode.split("_")[0])
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from offpolicy_amd import dist as opdist
        fast = opdist.setup_fast_allreduce(torch.device("cuda:0"))
        res = _rddpg_steps(name, (rank, world), dev_w=(mode == "auto_devprio"))
        torch.cuda.synchronize()
        bad = opdist._fast.timed_out() if fast else False
        out_q.put((rank, bool(fast), bad) + res)
    finally:
        torch.distributed.destroy_process_group()


def _rddpg_steps(name, shard, dev_w=False):
    """Two R_MATD3 updates (+ PER priorities) on the fixture's batch; shard = (rank, world) trains on that share. dev_w: the importance
    weights are a DEVICE tensor (device-resident trees): the priorities then stay in HBM and, at world > 1, are gathered inside the
    critic's gradient all-reduce (dist.priority_slots) instead of through a host all-gather."""
    from conftest import load_golden
    from golden_util import EP_KEYS
    import test_gpu_rddpg as R
    from offpolicy_amd import dist as opdist
    g = load_golden(name)
    dims, buf, policy, trainer = R.build(g)
    R.load_fixture_weights(g, policy, check=False)
    trainer.device_noise = False
    d = {k: {"policy_0": g["ep/" + k]} for k in EP_KEYS}
    buf.insert(len(g["idx_range"]), *[d[k] for k in EP_KEYS])
    inds = np.asarray(g["inds"])
    B = len(inds) - len(inds) % 2
    inds = inds[:B]
    w = (np.asarray(g["per_weights"])[:B] if "per_weights" in g else None)
    rank, world = shard
    per = B // world
    mine = slice(rank * per, (rank + 1) * per)
    T, N, A = dims.episode_length, dims.n_agents, dims.act_dim
    prios = []
    for st in range(2):
        # the reference's noise stream is laid out [T(+1), N*B, A] with row = agent*B + b: draw the FULL batch's noise and
        # hand each rank the columns of its episodes, so that sharded and full-batch runs consume identical numbers
        torch.manual_seed(1000 + st)
        u_t = torch.FloatTensor(T + 1, N, B, A).uniform_() if policy.target_noise is not None else None
        u_a = torch.FloatTensor(T, N, B, A).uniform_()
        trainer._noise_override = (None if u_t is None else u_t[:, :, mine].reshape(T + 1, N * per, A).contiguous(),
                                   u_a[:, :, mine].reshape(T, N * per, A).contiguous())
        s = buf.policy_buffers["policy_0"].sample_inds(inds[mine])
        wm = None if w is None else (torch.as_tensor(w[mine], dtype=torch.float32).cuda() if dev_w else w[mine])
        batch = tuple({"policy_0": a} for a in s) + (wm, inds if w is not None else None)
        info, prio, _ = trainer.shared_train_policy_on_batch("policy_0", batch)
        policy.soft_target_updates()
        if dev_w and prio is not None:
            # uniform contract: the call returns THIS rank's share; what the gradient all-reduce gathered (all ranks, in HBM, no host trip)
            # is kept in trainer.gathered_priorities, and the documented follow-up call works with or without it
            assert torch.is_tensor(prio) and prio.is_cuda and len(prio) == per, len(prio)
            have = trainer.gathered_priorities
            assert torch.is_tensor(have) and have.is_cuda and len(have) == B
            full = opdist.allgather_cat(prio, have=have)
            assert torch.equal(full, opdist.allgather_cat(prio)) and torch.equal(full[mine], prio)
            prios.append(full.cpu().numpy())
        else:
            prios.append(opdist.allgather_cat(prio))
    torch.cuda.synchronize()
    return (policy.critic._flat.cpu().numpy(), policy.actor._flat.cpu().numpy(), policy.target_critic._flat.cpu().numpy(),
            None if prios[0] is None else np.asarray(prios))


def _allreduce_world_worker(rank, world, port, out_q):
    """One-shot all-reduce between `world` processes that all sit on cuda:0: every rank maps the world - 1 other exchange buffers
    through HIP IPC and runs the same kernel a real node runs (there: one rank per GPU, the buffers reached over xGMI)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from offpolicy_amd.dist import OneShotAllreduce
        dev = torch.device("cuda:0")
        ar = OneShotAllreduce(dev, rank, world, max_floats=1 << 18, timeout_ms=20000)
        assert len(ar._mapped) == world - 1
        g = torch.Generator(device="cpu").manual_seed(4321 + rank)
        ok, digests = True, []
        # 8 x ceil(n / 1024) spinning workgroups share ONE GPU here: keep n below the co-residency limit (a node has a GPU per rank)
        for n in (118795, 5, 1024, 1 << 17, 118795, 33333):
            x = torch.randn(n, generator=g)
            ref = x.clone()
            torch.distributed.all_reduce(ref, op=torch.distributed.ReduceOp.SUM)          # gloo, host
            y = x.to(dev)
            ar(y)
            y = y.cpu()
            ok = ok and bool(torch.allclose(y, ref, rtol=1e-5, atol=1e-5))
            digests.append(y.numpy().tobytes())
        timed_out = ar.timed_out()
        import hashlib
        h = hashlib.sha256(b"".join(digests)).hexdigest()
        torch.distributed.barrier()
        # a peer that never arrives: rank 0 alone starts one more exchange with a short timeout. It must come back (bounded spin),
        # raise the status flag AND leave NaN in every chunk instead of a sum over stale slots.
        poisoned = None
        if rank == 0:
            ar.ctx.timeout_ms = 200
            z = torch.ones(5000, device=dev)
            ar(z)
            torch.cuda.synchronize()
            poisoned = bool(torch.isnan(z).all().item()) and ar.timed_out()
        torch.distributed.barrier()
        out_q.put((rank, ok, bool(timed_out), h, poisoned, ar.epoch))
        ar.close()
    finally:
        torch.distributed.destroy_process_group()


def _maddpg_graph_worker(rank, world, port, name, mode, out_q):
    """Three MADDPG updates at world = 2 (both ranks on cuda:0), eagerly and as replays of the captured graph."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OPE_ALLREDUCE=mode)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden
        import test_gpu_ddpg as D
        from offpolicy_amd import dist as opdist
        fast = opdist.setup_fast_allreduce(torch.device("cuda:0"))
        g = load_golden(name)
        B = len(g["inds"]) - len(g["inds"]) % world
        per = B // world
        out = {}
        for how in ("eager", "graph"):
            dims, buf, policy, trainer = D.build(g)
            trainer.device_noise = True
            lists = [np.random.RandomState(7 + s).choice(len(buf), B) for s in range(3)]
            if how == "graph":
                if not fast:
                    try:
                        trainer.make_graphed_step(buf, per)
                        out[how] = "captured without the one-shot exchange"
                    except NotImplementedError:
                        out[how] = None          # the documented refusal (RCCL fallback): nothing to compare
                    continue
                step = trainer.make_graphed_step(buf, per)
            infos = []
            for s in range(3):
                mine = lists[s][rank * per:(rank + 1) * per]
                if how == "graph":
                    info = step(mine)
                else:
                    smp = buf.policy_buffers["policy_0"].sample_inds(mine)
                    info, _, _ = trainer.shared_train_policy_on_batch("policy_0", tuple({"policy_0": a} for a in smp) + (None, None))
                    policy.soft_target_updates()
                infos.append([float(info["critic_loss"]), float(info["critic_grad_norm"])])
            torch.cuda.synchronize()
            out[how] = (policy.critic._flat.cpu().numpy(), policy.actor._flat.cpu().numpy(), policy.target_critic._flat.cpu().numpy(),
                        np.asarray(infos), policy.critic_optimizer.step_count)
        bad = opdist._fast.timed_out() if fast else False
        out_q.put((rank, bool(fast), bad, out["eager"], out["graph"]))
    finally:
        torch.distributed.destroy_process_group()


def test_two_rank_maddpg_graphed_step_equals_eager_distributed_step():
    """The captured MADDPG update with the gradient all-reduces INSIDE the graph (one-shot exchange, device-held epoch:
    ope_allreduce_flat_dev) replays what the eager two-rank step computes; the ranks stay bitwise identical."""
    res = _spawn(_maddpg_graph_worker, "maddpg_spread", "auto")
    fast0, bad0, e0, g0 = res[0]
    fast1, bad1, e1, g1 = res[1]
    assert fast0 == fast1 and not bad0 and not bad1
    if not fast0:
        assert g0 is None and g1 is None, "without the one-shot exchange the graphed step must refuse"
        pytest.skip("one-shot all-reduce not verified on this box: the graphed step refused, as documented")
    for q in range(3):
        assert np.array_equal(g0[q], g1[q]) and np.array_equal(e0[q], e1[q]), q      # replicas identical, both ways
        np.testing.assert_allclose(g0[q], e0[q], rtol=0, atol=2e-6)
    np.testing.assert_allclose(g0[3], e0[3], rtol=2e-5)
    assert g0[4] == e0[4] == 3


@pytest.mark.parametrize("world", [8, 3])
def test_one_shot_allreduce_eight_ranks_on_one_device(world):
    """The flag / slot / IPC-mapping logic of the one-shot all-reduce at the world size a node runs (8), before a node ever runs it:
    8 processes on cuda:0 (gloo rendezvous), 7 IPC-mapped peer buffers each, six exchanges of different lengths (both parities,
    ragged tails) equal to gloo's sum and BITWISE identical on all ranks (fixed rank-order sums), no timeout; and a 3-rank
    (non-power-of-two) variant. Then the failure path: a lone caller times out, reports it and poisons its vector with NaN."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_allreduce_world_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=900)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(res[r][0] for r in range(world)), "a rank's result differs from gloo's sum"
    assert not any(res[r][1] for r in range(world)), "a wait timed out during the regular exchanges"
    assert len({res[r][2] for r in range(world)}) == 1, "ranks hold different bits"
    assert res[0][3] is True, "the lone caller was not poisoned / flagged"
    assert res[1][4] == 6 and res[0][4] == 7


def _spawn(worker, name, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=worker, args=(r, 2, port, name, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        item = q.get(timeout=600)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _documented_per_flow_worker(rank, world, port, name, mode, out_q):
    """INTEGRATION.md "Data parallel": sample the rank's share, train, `dist.allgather_cat(new_priorities)`, update_priorities -- as written
    there, for QMix (host and device-resident importance weights) and MLP MADDPG."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OPE_ALLREDUCE="auto")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden
        from offpolicy_amd import dist as opdist
        opdist.setup_fast_allreduce(torch.device("cuda:0"))
        g = load_golden(name)
        dev_w = mode == "device_weights"
        if name.startswith("qmix"):
            from gpu_util import build_from_fixture
            dims, buf, policy, trainer = build_from_fixture(g)
            inds = np.asarray(g["inds"])[:4]
            w = np.asarray(g["per_weights"])[:4]
            mine = opdist.shard_indices(np.arange(4))
            smp = buf.policy_buffers["policy_0"].sample_inds(inds[mine])
            wm = torch.as_tensor(w[mine], dtype=torch.float32).cuda() if dev_w else w[mine]
            batch = tuple({"policy_0": a} for a in smp) + (wm, inds)
            info, prio, idxes = trainer.train_policy_on_batch(batch)
        else:
            import test_gpu_ddpg as D
            from test_mlp_oracle_golden import T_KEYS
            dims, buf, policy, trainer = D.build(g)
            inds = np.asarray(g["inds"])[:6]
            w = np.asarray(g["per_weights"])[:6]
            mine = opdist.shard_indices(np.arange(6))
            smp = buf.policy_buffers["policy_0"].sample_inds(inds[mine])
            batch = tuple({"policy_0": a} for a in smp) + (w[mine], inds)
            torch.manual_seed(1000)
            info, prio, idxes = trainer.shared_train_policy_on_batch("policy_0", batch)
        assert len(prio) == len(mine), (len(prio), len(mine))          # the LOCAL share, whatever the trainer
        full = opdist.allgather_cat(prio)                                # the documented call, no extra argument
        assert len(full) == len(inds) and len(idxes) == len(inds)
        have = getattr(trainer, "gathered_priorities", None)
        if have is not None:                                             # (device weights: gathered inside the gradient all-reduce)
            again = opdist.allgather_cat(prio, have=have)
            assert torch.equal(torch.as_tensor(again).cpu(), torch.as_tensor(full).cpu())
        torch.cuda.synchronize()
        full = full.cpu().numpy() if torch.is_tensor(full) else np.asarray(full)
        out_q.put((rank, full.astype(np.float64), have is not None))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name,mode", [("qmix_tiny_huber_per", "host_weights"), ("qmix_tiny_huber_per", "device_weights"),
                                       ("maddpg_small_huber_per", "host_weights")])
def test_two_rank_documented_priority_flow(name, mode):
    """every trainer returns its LOCAL priorities and `dist.allgather_cat(new_priorities)` rebuilds the global vector,
    identically on both ranks; where the trainer gathered them itself (`gathered_priorities`) the shortcut gives the same values."""
    res = _spawn(_documented_per_flow_worker, name, mode)
    (p0, had0), (p1, had1) = res[0], res[1]
    assert np.array_equal(p0, p1) and had0 == had1
    assert had0 == (mode == "device_weights")
    assert np.all(np.isfinite(p0)) and np.all(p0 > 0)


@pytest.mark.parametrize("mode", ["auto", "rccl"])
def test_two_rank_qmix_step_equals_full_batch_step(mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    from gpu_util import build_from_fixture, batch_from
    name = "qmix_tiny"
    res = _spawn(_qmix_worker, name, mode)
    fast0, bad0, th0, tg0, info0 = res[0]
    fast1, bad1, th1, tg1, info1 = res[1]
    assert fast0 == fast1 and not bad0 and not bad1
    if mode == "rccl":
        assert not fast0
    print("one-shot all-reduce verified on this box:", fast0)
    # ranks hold identical parameters after the redundant optimizer steps (fixed-order sums: bitwise)
    assert np.array_equal(th0, th1) and np.array_equal(tg0, tg1) and np.array_equal(info0, info1)
    g = load_golden(name)
    dims, buf, policy, trainer = build_from_fixture(g)
    inds = np.asarray(g["inds"])[:4]
    want = []
    for _ in range(2):
        info, _, _ = trainer.train_policy_on_batch(batch_from(buf, inds))
        trainer.soft_target_updates()
        want.append([float(info[k]) for k in ("loss", "grad_norm", "Q_tot")])
    np.testing.assert_allclose(info0, np.asarray(want), rtol=2e-5)
    np.testing.assert_allclose(th0, trainer.theta.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(tg0, trainer.theta_tgt.cpu().numpy(), rtol=0, atol=2e-6)


def test_two_rank_rmatd3_per_step_equals_full_batch_step():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    name = "rmatd3_odd_per"
    res = _spawn(_rddpg_worker, name, "auto")
    fast0, bad0, c0, a0, tc0, p0 = res[0]
    fast1, bad1, c1, a1, tc1, p1 = res[1]
    assert fast0 == fast1 and not bad0 and not bad1
    assert np.array_equal(c0, c1) and np.array_equal(a0, a1) and np.array_equal(tc0, tc1) and np.array_equal(p0, p1)
    wc, wa, wtc, wp = _rddpg_steps(name, (0, 1))
    np.testing.assert_allclose(c0, wc, rtol=0, atol=3e-6)
    np.testing.assert_allclose(a0, wa, rtol=0, atol=3e-6)
    np.testing.assert_allclose(tc0, wtc, rtol=0, atol=3e-6)
    np.testing.assert_allclose(p0, wp, rtol=2e-5)


def _qmix_graph_worker(rank, world, port, name, mode, out_q):
    """Three QMIX updates at world = 2 (both ranks on cuda:0), eagerly and as replays of the captured graph (all-reduce inside)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OPE_ALLREDUCE=mode)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden
        from gpu_util import build_from_fixture, batch_from
        from offpolicy_amd import dist as opdist
        fast = opdist.setup_fast_allreduce(torch.device("cuda:0"))
        g = load_golden(name)
        out = {}
        for how in ("eager", "graph"):
            dims, buf, policy, trainer = build_from_fixture(g)
            trainer.fuse_soft_update = True
            n = len(buf)
            lists = [np.random.RandomState(11 + s).choice(n, 4) for s in range(3)]
            if how == "graph":
                if not fast:
                    try:
                        trainer.make_graphed_step(buf, 2)
                        out[how] = "captured without the one-shot exchange"
                    except NotImplementedError:
                        out[how] = None
                    continue
                step = trainer.make_graphed_step(buf, 2)
            infos = []
            for s in range(3):
                mine = lists[s][rank * 2:(rank + 1) * 2]
                if how == "graph":
                    info = step(mine)
                else:
                    info, _, _ = trainer.train_policy_on_batch(batch_from(buf, mine))
                    trainer.soft_target_updates()
                infos.append([float(info[k]) for k in ("loss", "grad_norm", "Q_tot")])
            torch.cuda.synchronize()
            refused = False
            if how == "graph":           # once the exchange is retired the graph must refuse to replay (it points into the retired buffers)
                opdist.disable_fast_allreduce("test")
                try:
                    step(lists[0][rank * 2:(rank + 1) * 2])
                except RuntimeError:
                    refused = True
            out[how] = (trainer.theta.cpu().numpy(), trainer.theta_tgt.cpu().numpy(), np.asarray(infos), refused)
        out_q.put((rank, bool(fast), out["eager"], out["graph"]))
    finally:
        torch.distributed.destroy_process_group()


def test_two_rank_qmix_graphed_step_equals_eager_distributed_step():
    """QMix.make_graphed_step at world = 2: the gradient all-reduce is captured with the kernels (one-shot exchange, device-held epoch) and
    the replays reproduce the eager two-rank steps; after dist.disable_fast_allreduce the graph refuses to replay."""
    res = _spawn(_qmix_graph_worker, "qmix_tiny", "auto")
    fast0, e0, g0 = res[0]
    fast1, e1, g1 = res[1]
    assert fast0 == fast1
    if not fast0:
        assert g0 is None and g1 is None, "without the one-shot exchange the graphed step must refuse"
        pytest.skip("one-shot all-reduce not verified on this box: the graphed step refused, as documented")
    for q in range(2):
        assert np.array_equal(g0[q], g1[q]) and np.array_equal(e0[q], e1[q]), q      # replicas identical, both ways
        np.testing.assert_allclose(g0[q], e0[q], rtol=0, atol=2e-6)
    np.testing.assert_allclose(g0[2], e0[2], rtol=2e-5)
    assert g0[3] and g1[3], "a graph that captured a retired exchange must refuse to replay"


def test_two_rank_rmatd3_device_priorities_are_gathered_inside_the_gradient_allreduce():
    """Prioritized multi-process step with device-resident importance weights: every rank writes its per-episode priorities into its own
    slots behind the critic's gradient tail, the ONE all-reduce of the step sums the disjoint slots, and every rank holds all B priorities
    in HBM -- no host all-gather. Values equal the single-process full-batch step's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    name = "rmatd3_odd_per"
    res = _spawn(_rddpg_worker, name, "auto_devprio")
    fast0, bad0, c0, a0, tc0, p0 = res[0]
    fast1, bad1, c1, a1, tc1, p1 = res[1]
    assert fast0 == fast1 and not bad0 and not bad1
    assert np.array_equal(c0, c1) and np.array_equal(a0, a1) and np.array_equal(p0, p1)
    wc, wa, wtc, wp = _rddpg_steps(name, (0, 1))
    np.testing.assert_allclose(c0, wc, rtol=0, atol=3e-6)
    np.testing.assert_allclose(a0, wa, rtol=0, atol=3e-6)
    np.testing.assert_allclose(p0, wp, rtol=2e-5)
