"""Data-parallel plumbing: one process per GPU, gradients exchanged in ONE flat all-reduce (RCCL over xGMI).

The reference has no distributed training (its only collective, `average_gradients`, offpolicy/utils/util.py:148-153,
is dead code). The update shards over episodes: each rank back-propagates the UN-normalised loss sum of its share
of the sampled episodes, then a single SUM all-reduce of `[grads | loss_sum | mask_count | qtot_sum | 0]`
(~475 KB for QMIX 3s5z) makes every rank hold the global sums; normalisation by the global mask count, global-norm
clipping and Adam then run redundantly and identically on every rank.
"""
import numpy as np
import torch


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def world():
    if is_distributed():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def shard_indices(inds, rank=None, world_size=None):
    """Rank r's contiguous share of the sampled episode indices (B must divide evenly)."""
    if rank is None:
        rank, world_size = world()
    inds = np.asarray(inds)
    assert len(inds) % world_size == 0, "batch size must be a multiple of the number of ranks"
    per = len(inds) // world_size
    return inds[rank * per:(rank + 1) * per]


class OneShotAllreduce(object):
    """Host side of csrc/ope_allreduce.hip: every rank allocates one fine-grained exchange buffer, the HIP IPC handles
    travel through torch.distributed.all_gather_object, every rank maps its peers' buffers, and `__call__` is ONE kernel
    launch on the current stream (push to all peers over xGMI, flag, wait, fixed-rank-order sum). Also usable with
    world == 1 (no process group needed): the vector makes a round trip through the own slot."""
    MAX_FLOATS = 1 << 20      # 4 MiB slots: the BASELINE configs' flat gradients are < 0.6 MB, QMIX 3s5z with the wide
    #                           --use_global_all_local_state centralized state (S = 2 232) is 2.5 MB

    def __init__(self, device, rank=0, world_size=1, max_floats=None, group=None, timeout_ms=0):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        self.device = torch.device(device)
        self.rank, self.world = int(rank), int(world_size)
        self.max_floats = int(max_floats or self.MAX_FLOATS)
        self._mapped = []
        with torch.cuda.device(self.device):
            nbytes = int(_lib.lib.ope_allreduce_buffer_bytes(self.max_floats, self.world))
            if nbytes < 0:
                _lib.check(nbytes, "ope_allreduce_buffer_bytes")
            buf = C.c_void_p(0)
            rc_alloc = int(_lib.lib.ope_allreduce_alloc(nbytes, C.byref(buf)))
            if rc_alloc != 0 and self.world == 1:
                _lib.check(rc_alloc, "ope_allreduce_alloc")
            self._buf = buf
            handle = (C.c_ubyte * 64)()
            peers = [None] * self.world
            if self.world > 1:
                # every rank takes part in the handle exchange even if its own allocation or export failed (None), so that a local
                # failure cannot leave the other ranks stuck in the collective
                rc = _lib.lib.ope_allreduce_ipc_export(buf, handle) if rc_alloc == 0 else rc_alloc
                handles = [None] * self.world
                me = (bytes(handle), int(self.device.index if self.device.index is not None else torch.cuda.current_device())) if rc == 0 else None
                torch.distributed.all_gather_object(handles, me, group=group)
                if any(h is None for h in handles):
                    raise _lib.OpeError("exchange buffer allocation / hipIpcGetMemHandle failed on rank(s) %s" % [q for q, h in enumerate(handles) if h is None])
                for q, (h, peer_dev) in enumerate(handles):
                    if q == self.rank:
                        continue
                    _lib.check(_lib.lib.ope_allreduce_enable_peer(peer_dev), "ope_allreduce_enable_peer(%d)" % peer_dev)
                    m = C.c_void_p(0)
                    hb = (C.c_ubyte * 64).from_buffer_copy(h)
                    _lib.check(_lib.lib.ope_allreduce_ipc_import(hb, C.byref(m)), "ope_allreduce_ipc_import")
                    self._mapped.append(m)
                    peers[q] = m.value
            peers[self.rank] = buf.value
            self.ctx = _lib.AllreduceCtx()
            self.ctx.rank, self.ctx.world, self.ctx.max_floats = self.rank, self.world, self.max_floats
            self.ctx.timeout_ms = int(timeout_ms)
            for q in range(self.world):
                self.ctx.peer[q] = peers[q]
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
            # the call counter lives on the device ({epoch of the previous exchange, ticket}: ope_allreduce_flat_dev), so every
            # launch is identical and the exchange can be captured in a HIP graph (MADDPG.make_graphed_step at world > 1). It wraps
            # over the even cycle 1, 2, ..., 0xFFFFFFFE, 1, ... inside the kernel (parity picks the buffer half; 0 is never used).
            self.epoch_state = torch.zeros(2, dtype=torch.int32, device=self.device)

    graph_safe = True      # nothing of a launch changes from call to call

    @property
    def epoch(self):
        """Exchanges completed so far, modulo the wrap (reads the device counter: synchronises)."""
        return int(self.epoch_state[0].item())

    def __call__(self, flat):
        assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.device == self.device and flat.numel() <= self.max_floats
        self._lib.check(self._lib.lib.ope_allreduce_flat_dev(self._C.byref(self.ctx), self._lib.ptr(self.epoch_state), self._lib.ptr(flat),
                                                             flat.numel(), self._lib.ptr(self.status), self._lib.current_stream()),
                        "ope_allreduce_flat_dev")
        return flat

    def timed_out(self):
        """True if any call so far gave up waiting for a peer (synchronises the stream)."""
        return bool(int(self.status.item()) != 0)

    def close(self):
        torch.cuda.synchronize(self.device)
        for m in self._mapped:
            self._lib.lib.ope_allreduce_ipc_close(m)
        self._mapped = []
        if self._buf is not None:
            self._lib.lib.ope_allreduce_free(self._buf)
            self._buf = None


_fast = None          # OneShotAllreduce once setup_fast_allreduce() has verified it, else None (torch.distributed / RCCL)
_fast_note = "rccl (torch.distributed all_reduce)"
_fast_gen = 0         # bumped whenever the one-shot path is switched on or off: a captured HIP graph holds raw pointers into ITS exchange buffers
_graph_refs = 0       # captured graphs that reference the current one-shot exchange (note_graph_capture)
_retired = []         # exchanges that were disabled while a captured graph still pointed into them: kept mapped, never launched again


def fast_generation():
    """Identity of the all-reduce path in use. A graphed step records it at capture and refuses to replay once it has changed
    (its captured kernels would push into buffers that are no longer part of any exchange)."""
    return _fast_gen


def note_graph_capture():
    """A HIP graph has just captured launches of the one-shot exchange: its buffers must outlive the graph. Returns the generation to check
    before every replay."""
    global _graph_refs
    if _fast is not None:
        _graph_refs += 1
    return _fast_gen


def fast_allreduce_failed():
    """True if the one-shot all-reduce is active and any of its calls gave up waiting for a peer (results then invalid)."""
    return _fast is not None and _fast.timed_out()


def disable_fast_allreduce(reason="disabled"):
    """Back to torch.distributed's all_reduce (RCCL) for the rest of the run; call it on every rank."""
    global _fast, _fast_note, _fast_gen, _graph_refs
    if _fast is not None:
        ar, _fast = _fast, None
        _fast_gen += 1
        _fast_note = "rccl (one-shot xGMI path %s)" % reason
        if _graph_refs > 0:
            # captured graphs still hold raw pointers into this exchange's local and peer buffers: unmapping them would turn a stray
            # replay into writes to freed memory. Keep everything mapped for the life of the process (a few MB); the graphs themselves
            # refuse to replay from now on (fast_generation() changed), so the buffers are never used again either.
            _retired.append(ar)
            _graph_refs = 0
            _close_after_barrier(None)
        else:
            _close_after_barrier(ar)


def _close_after_barrier(ar, group=None):
    """Release a OneShotAllreduce that is no longer used: its uncached exchange buffer and the peers' IPC mappings. Every rank
    calls this at the same point; the barrier makes sure no peer is still pushing into the buffer being freed."""
    try:
        if is_distributed():      # (also when THIS rank has nothing to free: its peers wait in the same barrier)
            torch.distributed.barrier(group=group)
        if ar is not None:
            ar.close()
    except Exception:      # tearing down a broken fast path must not take the (working) RCCL path with it
        pass


def graph_safe_allreduce(n_floats=0):
    """True when `allreduce_flat_` of a float32 CUDA vector of `n_floats` elements may be captured in a HIP graph: the one-shot exchange
    is in use (its launches are identical from call to call) and the vector fits its slots; torch.distributed's all_reduce is kept out of
    captures (allreduce_flat_ raises rather than fall back to it while a stream is capturing)."""
    return _fast is not None and getattr(_fast, "graph_safe", False) and int(n_floats) <= _fast.max_floats


def allreduce_backend():
    return "one-shot xGMI push (ope_allreduce_flat)" if _fast is not None else _fast_note


def setup_fast_allreduce(device, group=None):
    """Try to switch `allreduce_flat_` to the one-shot xGMI kernel. It is only used if, on EVERY rank, three all-reduces
    of random vectors (two sizes) agree with torch.distributed's result and no wait timed out; otherwise the process group's
    all_reduce (RCCL) stays. OPE_ALLREDUCE=rccl skips the attempt, OPE_ALLREDUCE=oneshot makes a failure fatal."""
    global _fast, _fast_note
    import os
    import sys
    mode = os.environ.get("OPE_ALLREDUCE", "auto")
    if not is_distributed() or mode == "rccl":
        return False
    rank, world_size = world()
    ok, err, ar = 1, "", None
    try:
        ar = OneShotAllreduce(device, rank, world_size, group=group, timeout_ms=500)    # verification: fail fast
    except Exception as e:     # allocation / IPC not available on this system
        ok, err = 0, repr(e)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    for n in (118795, 5, 1 << 20):          # every rank runs the same collectives whatever its local state
        x = torch.randn(n, generator=g).to(device)
        ref = x.clone()
        torch.distributed.all_reduce(ref, op=torch.distributed.ReduceOp.SUM, group=group)
        if ar is not None:
            ar(x)
            if not torch.allclose(x, ref, rtol=1e-5, atol=1e-5):
                ok, err = 0, "mismatch vs torch.distributed at n=%d" % n
    if ar is not None and ar.timed_out():
        ok, err = 0, "peer flags timed out"
    t = torch.tensor([ok], dtype=torch.int32, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=group)
    if int(t.item()) == 1:
        global _fast_gen
        ar.ctx.timeout_ms = 0      # default (10 s) from here on: training ranks may be skewed by host work
        _fast = ar
        _fast_gen += 1
        return True
    _fast_note = "rccl (one-shot xGMI path not verified: %s)" % (err or "another rank failed")
    _close_after_barrier(ar, group)
    if rank == 0:
        print("[ope.dist] one-shot all-reduce disabled: %s" % (err or "another rank failed"), file=sys.stderr)
    if mode == "oneshot":
        raise RuntimeError("OPE_ALLREDUCE=oneshot but the one-shot all-reduce did not verify: " + err)
    return False


def priority_slots(flat, n_head, local):
    """Device all-gather riding on the gradient all-reduce: `flat` = [gradient and tail (n_head floats) | world x B slots]; this rank's B
    per-sample values go into ITS slots, the others are zeroed, and the SUM all-reduce of the whole vector that follows leaves the
    rank-ordered concatenation in flat[n_head:] on every rank (disjoint slots: x + 0 + ... + 0, exact). Returns that view."""
    rank, world_size = world()
    B = int(local.numel())
    ext = flat[n_head:n_head + world_size * B]
    ext.zero_()
    ext[rank * B:(rank + 1) * B].copy_(local)
    return ext


def allreduce_flat_(flat, group=None):
    """In-place SUM all-reduce of one flat tensor; a single collective per training step."""
    if is_distributed():
        if _fast is not None and group is None and flat.dtype == torch.float32 and flat.numel() <= _fast.max_floats and flat.is_cuda:
            return _fast(flat)
        if flat.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("allreduce_flat_: a HIP graph is being captured and this vector (%s, %d elements) cannot go through the one-shot "
                               "exchange (%s): torch.distributed's all_reduce must not be captured" % (flat.dtype, flat.numel(), allreduce_backend()))
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM, group=group)
    return flat


def allgather_cat(x, group=None, have=None):
    """Concatenation over ranks (rank order) of a per-sample vector: the per-episode priorities of a sharded prioritized
    batch, so that every rank updates its replica of the sum/min trees with the same B values.

    Contract (every trainer, every world size): `train_policy_on_batch` / `shared_train_policy_on_batch` return THIS RANK's
    priorities (length = its share of the batch); `allgather_cat(new_priorities)` is the documented next call and always works.
    `have`: a trainer that already holds the concatenation -- with device-resident importance weights at world > 1 the priorities
    ride on the gradient all-reduce (`priority_slots`) and the trainer keeps the result in `trainer.gathered_priorities` -- can be
    passed here to skip the extra collective: `allgather_cat(prio, have=trainer.gathered_priorities)` returns a copy of it (the
    view itself lives in the gradient vector and is overwritten by the next step)."""
    if not is_distributed() or x is None:
        return x
    if have is not None:
        _, ws = world()
        assert len(have) == ws * len(x), "gathered_priorities does not belong to this step"
        return have.clone() if torch.is_tensor(have) else np.array(have)
    _, world_size = world()
    if torch.is_tensor(x):
        out = [torch.empty_like(x) for _ in range(world_size)]
        torch.distributed.all_gather(out, x.contiguous(), group=group)
        return torch.cat(out)
    t = torch.as_tensor(np.asarray(x))
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.distributed.get_backend(group) == "nccl" else t.device
    t = t.to(dev)
    out = [torch.empty_like(t) for _ in range(world_size)]
    torch.distributed.all_gather(out, t, group=group)
    return torch.cat(out).cpu().numpy()
