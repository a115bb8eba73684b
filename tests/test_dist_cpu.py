"""CPU, world_size 2 over gloo: the data-parallel algebra and plumbing of off-policy_amd/dist.py.

Each rank back-propagates the UN-normalised loss sum of its half of the sampled episodes (here with the CPU oracle
standing in for the HIP step, which needs a GPU), the flat [grads | loss_sum | mask_count | qtot_sum] vector goes
through ONE all-reduce, and the result must equal the single-process full-batch quantities -- which is exactly what
QMix.train_policy_on_batch relies on when torch.distributed is initialised."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden
        from golden_util import oracle_from, reference_store_from
        from oracle import qmix_oracle as O
        from offpolicy_amd import dist as opdist
        g = load_golden(name)
        orc, dims = oracle_from(g)
        store, _ = reference_store_from(g)
        inds = np.asarray(g["inds"])[:4]
        assert opdist.is_distributed() and opdist.world() == (rank, world)
        mine = opdist.shard_indices(inds)
        assert len(mine) == len(inds) // world and np.array_equal(mine, inds[rank * 2:(rank + 1) * 2])
        grads, ls, cnt, qs = orc.loss_sum_and_grads(O.sample_inds(store, mine))
        flat = torch.cat([v.flatten() for v in grads.values()] + [torch.tensor([ls, cnt, qs, 0.0])])
        opdist.allreduce_flat_(flat)
        out_q.put((rank, flat.numpy()))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name", ["qmix_tiny"])
def test_sharded_gradients_allreduce_to_full_batch(name):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    from golden_util import oracle_from, reference_store_from
    from oracle import qmix_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0], res[1]), "ranks disagree after the all-reduce"
    g = load_golden(name)
    orc, _ = oracle_from(g)
    store, _ = reference_store_from(g)
    grads, ls, cnt, qs = orc.loss_sum_and_grads(O.sample_inds(store, np.asarray(g["inds"])[:4]))
    want = torch.cat([v.flatten() for v in grads.values()] + [torch.tensor([ls, cnt, qs, 0.0])]).numpy()
    np.testing.assert_allclose(res[0], want, rtol=2e-4, atol=1e-5)


def _world8_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from offpolicy_amd import dist as opdist
        B, n_head = 4, 12                       # global batch 32 over 8 ranks (BASELINE config 4's partition: 4 episodes per GPU)
        inds = np.random.RandomState(5).choice(1000, B * world)        # same draw on every rank
        mine = opdist.shard_indices(inds)
        assert np.array_equal(mine, inds[rank * B:(rank + 1) * B])
        uneven = None
        try:
            opdist.shard_indices(np.arange(B * world + 3))
        except AssertionError as e:
            uneven = str(e)
        # [gradient + tail | world x B priority slots]: the rank's share of the "gradient", its own priorities in its own slots
        flat = torch.zeros(n_head + B * world)
        flat[:n_head] = torch.arange(n_head, dtype=torch.float32) * (rank + 1)
        local = torch.as_tensor(mine, dtype=torch.float32) * 0.5 + 1.0
        view = opdist.priority_slots(flat, n_head, local)
        assert view.data_ptr() == flat[n_head:].data_ptr() and torch.count_nonzero(view) == B
        opdist.allreduce_flat_(flat)
        full = opdist.allgather_cat(local)                       # the documented call
        again = opdist.allgather_cat(local, have=view)           # the shortcut for what the all-reduce already gathered
        out_q.put((rank, flat.numpy(), full.numpy(), again.numpy(), uneven))
    finally:
        torch.distributed.destroy_process_group()


def test_world8_shares_priority_slots_and_uneven_batches():
    """Eight ranks over gloo (the node the driver scales to): contiguous shares of one global index draw, the ranks'
    priorities gathered INSIDE the gradient all-reduce through disjoint slots (exact: x + 0 + ... + 0), the same values from the
    documented `allgather_cat`, and a batch that does not divide over the ranks refused by `shard_indices`."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 977) % 2000)
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=300)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    inds = np.random.RandomState(5).choice(1000, 4 * world)
    want_prio = inds.astype(np.float32) * 0.5 + 1.0
    want_head = np.arange(12, dtype=np.float32) * sum(range(1, world + 1))
    for r in range(world):
        flat, full, again, uneven = res[r]
        np.testing.assert_array_equal(flat[:12], want_head)
        np.testing.assert_array_equal(flat[12:], want_prio)          # rank order = global batch order, bit-exact
        np.testing.assert_array_equal(full, want_prio)
        np.testing.assert_array_equal(again, want_prio)
        assert uneven and "multiple of the number of ranks" in uneven


def test_single_process_helpers_are_noops():
    from offpolicy_amd import dist as opdist
    assert not opdist.is_distributed() and opdist.world() == (0, 1)
    t = torch.arange(4.0)
    assert opdist.allreduce_flat_(t) is t and torch.equal(t, torch.arange(4.0))
    assert np.array_equal(opdist.shard_indices(np.arange(6), 1, 3), [2, 3])
    with pytest.raises(AssertionError):
        opdist.shard_indices(np.arange(5), 0, 2)


def test_disabling_the_one_shot_exchange_keeps_buffers_that_captured_graphs_point_into():
    """MADDPG.make_graphed_step at world > 1 captures launches of the one-shot exchange, i.e. raw pointers into its
    local and peer buffers. `disable_fast_allreduce` must then (a) NOT unmap those buffers, (b) change `fast_generation()` so that the
    graphed step refuses to replay; without a captured graph it frees them as before. Host logic only: a stand-in exchange object."""
    from offpolicy_amd import dist as D

    class Fake(object):
        max_floats, graph_safe, closed = 1000, True, 0

        def close(self):
            self.closed += 1

        def timed_out(self):
            return False
    saved = (D._fast, D._fast_note, D._fast_gen, D._graph_refs, list(D._retired))
    try:
        a = Fake()
        D._fast, D._graph_refs = a, 0
        assert D.graph_safe_allreduce(1000) and not D.graph_safe_allreduce(1001)      # slots must hold the vector
        gen = D.note_graph_capture()
        assert gen == D.fast_generation() and D._graph_refs == 1
        D.disable_fast_allreduce("test")
        assert D._fast is None and D.fast_generation() != gen and a.closed == 0 and a in D._retired and D._graph_refs == 0
        assert not D.graph_safe_allreduce(1)
        b = Fake()
        D._fast = b
        D.disable_fast_allreduce("test")            # no graph refers to this one: released
        assert b.closed == 1 and b not in D._retired
    finally:
        D._fast, D._fast_note, D._fast_gen, D._graph_refs = saved[:4]
        D._retired[:] = saved[4]
