"""Episode replay for recurrent policies, resident in HBM.

Drop-in mirror of offpolicy/utils/rec_buffer.py (`RecReplayBuffer` 10-82, `RecPolicyBuffer` 85-240,
`PrioritizedRecReplayBuffer` 243-324): same constructor arguments, `insert`, `sample`, `update_priorities`,
`__len__`, same 9-tuple of `{policy_id: array}` dicts. Differences by design:

* storage is EPISODE-major float32 on the GPU (`[capacity, T(+1), N, dim]`), written by `ope_store_insert`;
* `sample()` runs `ope_store_gather` and returns torch CUDA tensors shaped exactly like the reference's arrays
  (`[N, T(+1), B, dim]` views over `[T(+1), N, B, dim]` memory), which is also the `[T(+1), N*B, dim]` row-stacked
  layout the trainer kernels read -- no host round trip, no `torch.cat`. `.cpu().numpy()` on any of them gives the
  reference's array bit for bit.
* PER `insert` gives EVERY inserted slot the max priority (the reference only touches `range(idx[0], idx[1])`, which
  crashes for one episode and skips slots otherwise: SURVEY.md Appendix A-3).
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from .ring import RingIndex
from .segment_tree import SumSegmentTree, MinSegmentTree
from .spaces import get_dim_from_space

_INDS_MODE = os.environ.get("OPE_INDS_MODE", "args")    # args (kernel-argument block, B <= 512) | copy | zerocopy (RecPolicyBuffer._upload_inds)

_FIELD_ORDER = ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts")


def _shard(x, rank, world):
    """Rank `rank`'s contiguous share of a per-sample vector (numpy or tensor)."""
    n = len(x)
    assert n % world == 0, "the global batch must be a multiple of the number of ranks"
    per = n // world
    return x[rank * per:(rank + 1) * per]


def _shape_of(space):
    name = space.__class__.__name__
    if name == "Box":
        return tuple(space.shape)
    if name == "list":
        return tuple(space)
    raise NotImplementedError


class StoreObs(object):
    """The `obs` entry of a sampled batch whose rows were LEFT IN THE STORE (RecPolicyBuffer.lazy_obs; SURVEY.md section 8(d): "fused into
    the first consumer, written = 0"). The reference's sample_inds copies every field (rec_buffer.py:206-238) and the trainer reads obs
    back (qmix.py:108-109); the largest field by far (3s5z: 39 of 48 MB per batch) is read by exactly two kernels of the step, which can
    fetch the 4 * D-byte rows through the episode index instead. A trainer that can (QMix with ope_qmix_obs_ref_ok) takes this object
    as is; anything else calls `materialize()` -- or just uses it as an array / tensor (`shape`, `__array__`, `to`, `cpu`, indexing all
    materialize) -- and gets what sample_inds would have returned, [N, T+1, B, D]. Valid until the next insert() into the buffer: the
    rows are read at training time, so a later ring write would change the batch (checked: StaleBatchError)."""

    def __init__(self, buf, inds_dev, batch):
        self._buf, self.inds, self.batch = buf, inds_dev, int(batch)
        self._gen = buf._insert_gen
        self._dense = None
        d = buf.dims
        self.shape = (d.n_agents, d.episode_length + 1, self.batch, d.obs_dim)
        self.dtype, self.device = torch.float32, buf.device

    def check_fresh(self):
        if self._gen != self._buf._insert_gen:
            raise StaleBatchError("this batch left its observations in the replay store, and the store has been written since it was "
                                  "sampled: materialize() a batch you keep across insert(), or sample with lazy_obs off")

    def ref(self):
        """ope_obs_ref of the rows (the caller keeps this object alive until the step has run)."""
        self.check_fresh()
        r = _lib.ObsRef()
        r.store_obs, r.inds, r.capacity = _lib.ptr(self._buf.obs).value, _lib.ptr(self.inds).value, self._buf.buffer_size
        return r

    def materialize(self):
        """The gathered tensor, as sample_inds returns it with lazy_obs off ([N, T+1, B, D] view of a [T+1, N, B, D] batch)."""
        if self._dense is None:
            self.check_fresh()
            b, d = self._buf, self._buf.dims
            out = torch.empty((d.episode_length + 1, d.n_agents, self.batch, d.obs_dim), dtype=torch.float32, device=b.device)
            _lib.check(_lib.lib.ope_store_gather(C.byref(d), b.buffer_size, C.byref(b._obs_slot(b.obs)), _lib.ptr(self.inds), self.batch,
                                                 C.byref(b._obs_slot(out)), _lib.ptr(b._bad_index), _lib.current_stream()), "ope_store_gather(obs)")
            self._dense = out.permute(1, 0, 2, 3)
        return self._dense

    def __array__(self, dtype=None, copy=None):
        a = self.materialize().cpu().numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, k):
        return self.materialize()[k]

    def __len__(self):
        return self.shape[0]

    def to(self, *a, **k):
        return self.materialize().to(*a, **k)

    def cpu(self):
        return self.materialize().cpu()


class StaleBatchError(RuntimeError):
    pass


class RecPolicyBuffer(object):
    def __init__(self, buffer_size, episode_length, num_agents, obs_space, share_obs_space, act_space,
                 use_same_share_obs, use_avail_acts, use_reward_normalization=False, device=None, _reward_mask=True):
        self.buffer_size = int(buffer_size)
        self.episode_length = int(episode_length)
        self.num_agents = int(num_agents)
        self.use_same_share_obs = use_same_share_obs
        self.use_avail_acts = use_avail_acts
        self.use_reward_normalization = use_reward_normalization
        # normalised to an indexed device ("cuda" -> "cuda:0") so that device comparisons with tensors hold
        self.device = torch.empty(0, device=torch.device(device if device is not None else "cuda:0")).device
        self._bad_index = torch.zeros(1, dtype=torch.int32, device=self.device)   # set by the kernels on an out-of-range index
        self.gather_tune = None     # this store's gather knobs: dict of ope_gather_tune fields (include/ope.h); None = the process defaults
        self.lazy_obs = False       # True: sample_inds leaves the observation rows in the store (returns a StoreObs for `obs`)
        self._insert_gen = 0        # ring writes so far (a StoreObs is valid for the count it was sampled at)
        self._reward_mask = bool(_reward_mask)      # episodes: skip steps after the episode end; transitions: plain mean/std
        self._stats_dirty = True
        self._ring = RingIndex(self.buffer_size)
        obs_dim = _shape_of(obs_space)[0]
        share_dim = _shape_of(share_obs_space)[0]
        act_dim = int(np.sum(get_dim_from_space(act_space)))
        T, cap, N = self.episode_length, self.buffer_size, self.num_agents
        self.dims = _lib.Dims(N, act_dim, obs_dim, share_dim, T)
        z = dict(dtype=torch.float32, device=self.device)
        # same initial values as rec_buffer.py:120-141 (avail/dones/dones_env default to ones)
        self.obs = torch.zeros((cap, T + 1, N, obs_dim), **z)
        # shared centralized observation [cap, T+1, S], or one per agent [cap, T+1, N, S] (rec_buffer.py:120-125). The
        # per-agent form rides the kernels' agent-indexed `obs` slot in a second launch (dims with obs_dim = S): no ABI change.
        self.share_obs = torch.zeros((cap, T + 1, share_dim) if use_same_share_obs else (cap, T + 1, N, share_dim), **z)
        self._share_dims = _lib.Dims(N, act_dim, share_dim, share_dim, T)
        self.acts = torch.zeros((cap, T, N, act_dim), **z)
        self.avail_acts = torch.ones((cap, T + 1, N, act_dim), **z) if use_avail_acts else None
        self.rewards = torch.zeros((cap, T, N, 1), **z)
        self.dones = torch.ones((cap, T, N, 1), **z)
        self.dones_env = torch.ones((cap, T, 1), **z)
        if use_reward_normalization:
            self._stats = torch.zeros(4, **z)
            self._stats_scratch = torch.empty(int(_lib.lib.ope_reward_stats_scratch_bytes()), dtype=torch.uint8, device=self.device)

    def _upload_inds(self, inds):
        """Episode indices -> device without stalling the host: a pageable `.to(device)` is a synchronous copy that makes
        the host wait for all queued GPU work every step. Instead the indices go through a small ring of pinned staging
        buffers with asynchronous copies into a ring of device tensors. OPE_INDS_MODE=zerocopy skips the copy and lets the
        gather read the pinned buffer itself (hipHostMalloc memory is mapped into the device's address space): the step
        loses the blit kernel and its barrier (~1 % faster at 3s5z, ~6 % for the eager MLP-MADDPG step), the gather pays
        the host-link latency (~6 % slower), so it is opt-in. A copy on a side stream was measured too: no gain, the
        cross-stream wait costs what the barrier did. A slot is reused only after the last launch that read it has
        completed (`_release_inds`)."""
        B = int(inds.shape[0])
        ring = getattr(self, "_ind_ring", None)
        if ring is None or ring[0][0].numel() < B:
            n = max(B, 256)
            ring = [(torch.empty(n, dtype=torch.int64).pin_memory(), torch.cuda.Event(),
                     torch.empty(n, dtype=torch.int64, device=self.device)) for _ in range(8)]
            self._ind_ring, self._ind_slot, self._ind_used = ring, 0, [False] * 8
        k = self._ind_slot
        self._ind_slot = (k + 1) % len(ring)
        host, consumed, dev = ring[k]
        if self._ind_used[k]:
            consumed.synchronize()
        host[:B].copy_(torch.from_numpy(inds))
        self._ind_last = k
        if _INDS_MODE == "zerocopy":
            return host[:B]
        dev[:B].copy_(host[:B], non_blocking=True)
        return dev[:B]

    def _release_inds(self):
        """Mark the point on the current stream after which the last staged index slot may be overwritten."""
        k = getattr(self, "_ind_last", None)
        if k is not None:
            self._ind_ring[k][1].record()
            self._ind_used[k] = True

    def reward_stats(self):
        """Device tensor [mean, std, count, 0] of the rewards currently stored (rec_buffer.py:209-220); recomputed after
        every insert, on the device, without a host round trip."""
        if self._stats_dirty:
            _lib.check(_lib.lib.ope_store_reward_stats(C.byref(self.dims), self.filled_i, _lib.ptr(self.rewards),
                                                       _lib.ptr(self.dones_env) if self._reward_mask else None,
                                                       _lib.ptr(self._stats_scratch), _lib.ptr(self._stats), _lib.current_stream()),
                       "ope_store_reward_stats")
            self._stats_dirty = False
        return self._stats

    # reference attribute names
    @property
    def filled_i(self):
        return self._ring.filled_i

    @property
    def current_i(self):
        return self._ring.current_i

    def __len__(self):
        return self._ring.filled_i

    def _fields(self, tensors):
        f = _lib.Fields()
        for k in _FIELD_ORDER:
            setattr(f, k, _lib.ptr(tensors.get(k)).value)
        return f

    def _store_fields(self):
        # (the rings are allocated once: their pointers are read once; a copy is returned because callers add fields to it)
        f = getattr(self, "_store_fields_cache", None)
        if f is None or f.obs != self.obs.data_ptr():
            f = self._store_fields_cache = self._fields(dict(obs=self.obs, share_obs=self.share_obs if self.use_same_share_obs else None, acts=self.acts,
                                                             rewards=self.rewards, dones=self.dones, dones_env=self.dones_env, avail_acts=self.avail_acts))
        g = _lib.Fields()
        C.memmove(C.byref(g), C.byref(f), C.sizeof(g))
        return g

    def _obs_slot(self, t):
        """Fields block with only the agent-indexed `obs` slot set (per-agent centralized observations use it with obs_dim = S)."""
        f = _lib.Fields()
        f.obs = _lib.ptr(t).value
        return f

    def insert(self, num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts=None):
        """Ring write of `num_insert_episodes` episodes given in the reference's time-major layout
        ([T(+1), n, N, dim]); returns idx_range like rec_buffer.py:146-190."""
        acts = np.asarray(acts)
        assert acts.shape[0] == self.episode_length, ("different dimension!")
        n = int(num_insert_episodes)
        idx_range = self._ring.next_slots(n)
        share_obs = np.asarray(share_obs)
        if self.use_same_share_obs and share_obs.ndim == 4:
            share_obs = share_obs[:, :, 0]          # all agents share the centralized observation (rec_buffer.py:176-177)
        host = dict(obs=obs, acts=acts, rewards=rewards, dones=dones, dones_env=dones_env)
        if self.use_same_share_obs:
            host["share_obs"] = share_obs
        if self.use_avail_acts:
            host["avail_acts"] = avail_acts
        staged = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v), dtype=np.float32)).to(self.device, non_blocking=False)
                  for k, v in host.items()}
        slots = torch.from_numpy(np.asarray(idx_range, dtype=np.int64)).to(self.device)
        sf, df = self._fields(staged), self._store_fields()
        if not self.use_avail_acts:
            sf.avail_acts = None
        _lib.check(_lib.lib.ope_store_insert(C.byref(self.dims), self.buffer_size, C.byref(df), C.byref(sf),
                                             _lib.ptr(slots), n, _lib.ptr(self._bad_index), _lib.current_stream()), "ope_store_insert")
        if not self.use_same_share_obs:   # [T+1, n, N, S] -> the per-agent ring, through the obs slot
            assert share_obs.ndim == 4, "per-agent centralized observations expected ([T+1, n, N, S])"
            staged["share_obs"] = torch.from_numpy(np.ascontiguousarray(share_obs, dtype=np.float32)).to(self.device)
            _lib.check(_lib.lib.ope_store_insert(C.byref(self._share_dims), self.buffer_size, C.byref(self._obs_slot(self.share_obs)),
                                                 C.byref(self._obs_slot(staged["share_obs"])), _lib.ptr(slots), n, _lib.ptr(self._bad_index),
                                                 _lib.current_stream()), "ope_store_insert(share_obs)")
        self._keepalive = (staged, slots)   # until the stream has consumed them
        self._stats_dirty = True
        self._insert_gen += 1
        if getattr(self, "_filled_dev", None) is not None:
            self._filled_device()            # captured device-sampling graphs see the new slots from their next replay on
        return idx_range

    def check_indices(self):
        """Host-array indices are range-checked before upload (the reference's numpy indexing raises IndexError there);
        DEVICE index tensors (device PER trees, graph replays) cannot be checked without a sync, so the kernels skip the rows
        of an out-of-range index and raise a flag in HBM. This reads the flag (one sync) and raises like numpy would."""
        if int(self._bad_index.item()) != 0:
            self._bad_index.zero_()
            raise IndexError("an episode index passed to sample_inds()/insert() was outside [0, %d)" % self.buffer_size)

    def alloc_batch(self, batch_size, obs=True):
        """Destination tensors of one gather of `batch_size` episodes, in the kernels' [T(+1), N, B, dim] layout (`obs=False`: without
        the observation tensor -- lazy_obs)."""
        d, B = self.dims, int(batch_size)
        T, N = d.episode_length, d.n_agents
        e = dict(dtype=torch.float32, device=self.device)
        out = dict(obs=torch.empty((T + 1, N, B, d.obs_dim), **e) if obs else None,
                   share_obs=torch.empty((T + 1, B, d.state_dim) if self.use_same_share_obs else (T + 1, N, B, d.state_dim), **e),
                   acts=torch.empty((T, N, B, d.act_dim), **e), rewards=torch.empty((T, N, B, 1), **e),
                   dones=torch.empty((T, N, B, 1), **e), dones_env=torch.empty((T, B, 1), **e))
        if self.use_avail_acts:
            out["avail_acts"] = torch.empty((T + 1, N, B, d.act_dim), **e)
        return out

    def sample_device(self, batch_size, seed, counter=None, out=None, extra=None):
        """sample(batch_size) drawn on the device: uniform over the filled slots with replacement, like the reference's
        np.random.choice(filled, batch_size) (rec_buffer.py:86), but from a Philox stream keyed by (`seed`, batch position,
        the device int32 `counter`[0]) inside the gather kernel -- no index upload, no host work, and a captured HIP graph
        replays with fresh indices as the counter advances. Returns (7-tuple as sample_inds, int64 device tensor of the drawn
        indices)."""
        inds = torch.empty(int(batch_size), dtype=torch.int64, device=self.device)
        return self.sample_inds(inds, out=out, extra=extra, _sampler=(int(seed), counter, self._filled_device())), inds

    def _filled_device(self):
        """DEVICE int32 [1] = number of filled slots, refreshed by every insert (stream-ordered): the device-sampling gather reads
        it at run time, so a captured HIP graph of sample_device() keeps drawing from everything inserted since its capture."""
        t = getattr(self, "_filled_dev", None)
        if t is None:
            t = self._filled_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._filled_dev_value = -1
        if self._filled_dev_value != int(self.filled_i):
            self._filled_dev_value = int(self.filled_i)
            t.fill_(self._filled_dev_value)
        return t

    def sample_inds(self, sample_inds, timing_events=None, out=None, extra=None, _sampler=None, lazy_obs=None, live_for=None, live_only=False,
                    live_region=0):
        """Gather the given episode slots; same 7-tuple as rec_buffer.py:192-240 (CUDA tensors, reference shapes).
        `timing_events`: optional (start, end) torch.cuda.Event pair recorded tightly around the gather launch.
        `out`: optional destination from `alloc_batch` (HIP-graph replays read the batch from fixed addresses); the
        default is a fresh batch per call, like the reference's fancy-index copy.
        `extra`: optional (store, out) pair of [cap, T, N, 1] / [T, N, B, 1] tensors copied by the same launch (the transition
        buffers' valid_transition flag).
        `lazy_obs` (default: self.lazy_obs): leave the observation rows in the store and return a StoreObs in their place.
        `live_for`: the trainer this batch is sampled for (QMix). Where its step runs on live rows (ope_qmix_cfg.live_rows: only the rows
        before each episode's termination are computed, qmix.py:161-166), THIS gather launch also builds that step's row plan -- a few
        extra workgroups in front of the copy's, reading the store's dones_env of the sampled episodes (ope_store_gather_attach_live) --
        and the batch is tagged so that `trainer.train_policy_on_batch` skips its own plan launch in front of the step (~9 us of a 0.3 ms
        step). The tag is good for the NEXT train call on this batch; any other use of the batch is unaffected (the arrays are what the
        reference returns).
        `live_only` (with `live_for`; default off): the copy itself stops at each episode's termination -- of obs and share_obs (90 % of the
        batch's bytes) only the time entries t < len_b (share_obs: t <= len_b) are moved, the ones the live-row step reads; the later entries of those two arrays
        keep whatever the freshly allocated batch held. Every such (t, b) is multiplied by a zero mask in the reference's loss
        (qmix.py:161-166), and `live_for.train_policy_on_batch` refuses the batch if its step would not run on this plan's live rows. Use
        it when the batch goes straight into that call (bench.py does); leave it off to get the reference's arrays, padding included."""
        lazy = bool(self.lazy_obs if lazy_obs is None else lazy_obs)
        host_inds = None
        if torch.is_tensor(sample_inds):     # indices already on the device (HIP-graph replays keep them in a static tensor)
            assert sample_inds.dtype == torch.int64 and sample_inds.device == self.device, (sample_inds.dtype, sample_inds.device)
            dev_inds, B = sample_inds.contiguous(), int(sample_inds.shape[0])
            if lazy and _sampler is None:      # (device sampling: the kernel WRITES the drawn slots into this tensor, which sample_device allocated for the call)
                # a StoreObs keeps its indices past this call (the step reads the rows through them): it gets its OWN copy, so that a
                # caller that rewrites its index tensor before the step runs cannot redirect the rows (the graphed steps pass
                # lazy_obs=False: their kernels read the static tensor on purpose)
                dev_inds = dev_inds.clone()
        else:
            inds = np.ascontiguousarray(np.asarray(sample_inds, dtype=np.int64))
            B = int(inds.shape[0])
            if B and (inds.min() < -self.buffer_size or inds.max() >= self.buffer_size):
                raise IndexError("index out of bounds for a buffer of %d episodes" % self.buffer_size)   # as numpy fancy indexing
            if B and inds.min() < 0:
                inds = np.where(inds < 0, inds + self.buffer_size, inds)
            if B <= 512 and _INDS_MODE != "copy":
                host_inds = inds            # travel inside the kernel-argument block: no upload at all
            else:
                dev_inds = self._upload_inds(inds)
                if lazy:
                    dev_inds = dev_inds.clone()      # the staging ring is reused; a StoreObs keeps its indices
        d = self.dims
        fresh = out is None
        if fresh:
            # a fresh batch per call, as the reference's fancy-index copy -- taken from the spare that the previous call allocated BEHIND
            # its gather launch (seven torch.empty calls are ~35 us of host time: in front of the launch they are time the GPU idles
            # whenever the host is not running ahead, e.g. at the first step after a synchronize)
            spare = getattr(self, "_spare_batch", None)
            self._spare_batch = None
            skey = (B, not lazy, _lib.current_stream().value)      # (the caching allocator ties a block to the stream it was allocated on)
            out = spare[1] if (spare is not None and spare[0] == skey) else self.alloc_batch(B, obs=not lazy)
        else:
            assert out["acts"].shape[2] == B, "destination batch does not match the number of indices"
        if lazy:
            out = dict(out, obs=None)
        of, sf = self._fields(out if self.use_same_share_obs else {k: v for k, v in out.items() if k != "share_obs"}), self._store_fields()
        ref_inds = None
        if lazy and (host_inds is not None):      # the consumers read the slots from HBM: the gather writes them out beside its copies
            ref_inds = torch.empty(B, dtype=torch.int64, device=self.device)
        if extra is not None:
            sf.valid_transition, of.valid_transition = _lib.ptr(extra[0]).value, _lib.ptr(extra[1]).value
        live_tag = None
        if (live_for is not None and getattr(live_for, "build_live_plan", None) is not None and self.use_same_share_obs and _sampler is None and
                not (lazy and host_inds is not None)):      # (the attachment rides on the FIRST gather launch below: the plain forms)
            live_tag = (live_for.build_live_plan(self, host_inds, B, live_only=bool(live_only), region=int(live_region)) if (live_only or live_region) else
                        live_for.build_live_plan(self, host_inds, B))
        if timing_events is not None:
            timing_events[0].record()
        if _sampler is not None:
            seed, counter, filled_dev = _sampler
            assert self.use_same_share_obs, "device sampling: shared centralized observations only"
            assert int(self.filled_i) >= 1, "sample_device on an empty buffer"
            _lib.check(_lib.lib.ope_store_gather_sampled(C.byref(d), self.buffer_size, int(self.filled_i), _lib.ptr(filled_dev), C.byref(sf), seed, _lib.ptr(counter), B,
                                                         C.byref(of), _lib.ptr(dev_inds), _lib.current_stream()), "ope_store_gather_sampled")
        elif ref_inds is not None:
            tune = _lib.GatherTune(**{k: int(v) for k, v in (self.gather_tune or {}).items()})
            _lib.check(_lib.lib.ope_store_gather_ref(C.byref(d), self.buffer_size, C.byref(sf), None, host_inds.ctypes.data_as(C.c_void_p), B, C.byref(of),
                                                     _lib.ptr(ref_inds), _lib.ptr(self._bad_index), C.byref(tune), _lib.current_stream()), "ope_store_gather_ref")
        elif self.gather_tune is not None:      # this store's own gather knobs (A/B sweeps; the process defaults otherwise): ope_gather_tune
            tune = _lib.GatherTune(**{k: int(v) for k, v in self.gather_tune.items()})
            _lib.check(_lib.lib.ope_store_gather_tuned(C.byref(d), self.buffer_size, C.byref(sf), None if host_inds is not None else _lib.ptr(dev_inds),
                                                       host_inds.ctypes.data_as(C.c_void_p) if host_inds is not None else None, B, C.byref(of),
                                                       _lib.ptr(self._bad_index), C.byref(tune), _lib.current_stream()), "ope_store_gather_tuned")
        elif host_inds is not None:
            _lib.check(_lib.lib.ope_store_gather_host_inds(C.byref(d), self.buffer_size, C.byref(sf), host_inds.ctypes.data_as(C.c_void_p), B,
                                                           C.byref(of), _lib.current_stream()), "ope_store_gather_host_inds")
        else:
            _lib.check(_lib.lib.ope_store_gather(C.byref(d), self.buffer_size, C.byref(sf), _lib.ptr(dev_inds), B,
                                                 C.byref(of), _lib.ptr(self._bad_index), _lib.current_stream()), "ope_store_gather")
        if not self.use_same_share_obs:   # per-agent centralized observations: second gather through the obs slot
            so, ss = self._obs_slot(out["share_obs"]), self._obs_slot(self.share_obs)
            if host_inds is not None:
                _lib.check(_lib.lib.ope_store_gather_host_inds(C.byref(self._share_dims), self.buffer_size, C.byref(ss),
                                                               host_inds.ctypes.data_as(C.c_void_p), B, C.byref(so), _lib.current_stream()),
                           "ope_store_gather_host_inds(share_obs)")
            else:
                _lib.check(_lib.lib.ope_store_gather(C.byref(self._share_dims), self.buffer_size, C.byref(ss), _lib.ptr(dev_inds), B,
                                                     C.byref(so), _lib.ptr(self._bad_index), _lib.current_stream()), "ope_store_gather(share_obs)")
        if timing_events is not None:
            timing_events[1].record()
        if fresh:
            self._spare_batch = (skey, self.alloc_batch(B, obs=not lazy))      # the next call's destination (never handed out twice)
        if host_inds is None and not torch.is_tensor(sample_inds):
            self._release_inds()
        if self.use_reward_normalization:
            _lib.check(_lib.lib.ope_reward_normalize(_lib.ptr(out["rewards"]), out["rewards"].numel(), _lib.ptr(self.reward_stats()),
                                                     _lib.current_stream()), "ope_reward_normalize")
        cast = lambda x: x.permute(1, 0, 2, 3)      # [N, T(+1), B, dim] view, as the reference's _cast
        out["dones_env"]._ope_live = live_tag       # (None: no plan rides with this batch)
        if lazy:
            obs_entry = StoreObs(self, ref_inds if ref_inds is not None else dev_inds, B)
            return (obs_entry, out["share_obs"] if self.use_same_share_obs else cast(out["share_obs"]), cast(out["acts"]), cast(out["rewards"]),
                    cast(out["dones"]), out["dones_env"], cast(out["avail_acts"]) if self.use_avail_acts else None)
        return (cast(out["obs"]), out["share_obs"] if self.use_same_share_obs else cast(out["share_obs"]), cast(out["acts"]),
                cast(out["rewards"]), cast(out["dones"]),
                out["dones_env"], cast(out["avail_acts"]) if self.use_avail_acts else None)

    def sample_inds_ahead(self, sample_inds, live_for=None, live_only=False, after=None, **kw):
        """The gather of `sample_inds` launched NOW on this buffer's side stream, to be consumed later: returns a handle whose `.get()` makes
        the (then) current stream wait for the copy and returns the 7-tuple of sample_inds. For a loop of consecutive updates with no insert
        in between (several train steps per collected episode):
            cur = buf.sample_inds_ahead(inds_0, live_for=trainer)
            for k in ...:
                batch = cur.get(); mid = buf.midstep_event(at=3)          # optional: an event the step records in its middle (ope_qmix_signal_event)
                trainer.train_policy_on_batch(batch)
                cur = buf.sample_inds_ahead(inds_{k+1}, live_for=trainer, after=mid)
        the HBM-bound copy of batch k + 1 then runs beside step k's latency-bound kernels (the scan's adjoint, the optimizer tail) instead of
        in front of step k + 1: same indices, same batches, same arithmetic, in the same order. `after`: the event the side stream waits for
        (default: everything enqueued on the current stream so far). The batch is written into one of TWO destination batches this buffer
        keeps for the purpose, alternately (no allocation, no allocator events on the stream): a handle's arrays are overwritten by the
        second sample_inds_ahead call after its own -- and the live plans alternate between the workspace's two regions the same way, so
        `after` must not fire before the step that consumed the handle two calls back has finished (the default and a mid-step event of the
        step just enqueued both satisfy that). A batch sampled ahead does not see episodes inserted after this call."""
        main = torch.cuda.current_stream(self.device)
        st = getattr(self, "_ahead", None)
        B = int(len(sample_inds))
        if st is None or st["B"] != B:
            st = self._ahead = {"B": B, "stream": torch.cuda.Stream(device=self.device), "n": 0,
                                "out": [self.alloc_batch(B), self.alloc_batch(B)]}
            torch.cuda.current_stream(self.device).synchronize()      # (the two batches exist before the side stream first writes one)
        st["n"] += 1
        slot = st["n"] & 1
        side = st["stream"]
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            s = self.sample_inds(sample_inds, live_for=live_for, live_only=live_only, live_region=slot, out=st["out"][slot], **kw)
            done = torch.cuda.Event()
            done.record(side)
        return _AheadBatch(s, done, self.device)

    def midstep_event(self, at=3):
        """An event the NEXT QMIX step launched from this thread records in front of its launch `at` (ope.h: ope_qmix_signal_event: 1 the GRU
        scan, 2 the (t, b)-row chain, 3 the scan's adjoint, 4 the weight gradients, 5 their reduction); pass it to sample_inds_ahead(after=)
        AFTER the train call has returned."""
        ring = getattr(self, "_mid_events", None)
        if ring is None:
            ring = self._mid_events = {"n": 0, "ev": [torch.cuda.Event(), torch.cuda.Event()]}
            for e in ring["ev"]:
                e.record(torch.cuda.current_stream(self.device))      # (creates the hipEvent_t; every step records it again)
        ring["n"] += 1
        ev = ring["ev"][ring["n"] & 1]
        _lib.check(_lib.lib.ope_qmix_signal_event(C.c_void_p(ev.cuda_event), int(at)), "ope_qmix_signal_event")
        return ev


class _AheadBatch(object):
    """A batch whose gather is in flight on the buffer's side stream (RecPolicyBuffer.sample_inds_ahead)."""
    def __init__(self, batch, done, device):
        self._batch, self._done, self._device = batch, done, device

    def get(self):
        torch.cuda.current_stream(self._device).wait_event(self._done)
        return self._batch


class RecReplayBuffer(object):
    def __init__(self, policy_info, policy_agents, buffer_size, episode_length, use_same_share_obs, use_avail_acts,
                 use_reward_normalization=False, device=None):
        self.policy_info = policy_info
        self.policy_buffers = {p_id: RecPolicyBuffer(buffer_size, episode_length, len(policy_agents[p_id]),
                                                     self.policy_info[p_id]['obs_space'],
                                                     self.policy_info[p_id]['share_obs_space'],
                                                     self.policy_info[p_id]['act_space'],
                                                     use_same_share_obs, use_avail_acts, use_reward_normalization,
                                                     device=device)
                               for p_id in self.policy_info.keys()}

    def __len__(self):
        return self.policy_buffers['policy_0'].filled_i

    def insert(self, num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts):
        for p_id in self.policy_info.keys():
            idx_range = self.policy_buffers[p_id].insert(num_insert_episodes, np.array(obs[p_id]),
                                                         np.array(share_obs[p_id]), np.array(acts[p_id]),
                                                         np.array(rewards[p_id]), np.array(dones[p_id]),
                                                         np.array(dones_env[p_id]), np.array(avail_acts[p_id]))
        return idx_range

    def _gather(self, inds):
        keys = ({}, {}, {}, {}, {}, {}, {})
        for p_id in self.policy_info.keys():
            for dst, val in zip(keys, self.policy_buffers[p_id].sample_inds(inds)):
                dst[p_id] = val
        return keys

    def sample(self, batch_size, shard=None):
        """Uniform sampling WITH replacement from the global numpy RNG, as rec_buffer.py:76.
        Data-parallel use (no reference counterpart): with `shard=(rank, world)` every rank draws the SAME `batch_size`
        global indices (equal seeds, equal store replicas) and gathers only its contiguous share of them."""
        inds = np.random.choice(self.__len__(), batch_size)
        if shard is not None:
            inds = _shard(inds, *shard)
        return self._gather(inds) + (None, None)


class PrioritizedRecReplayBuffer(RecReplayBuffer):
    def __init__(self, alpha, policy_info, policy_agents, buffer_size, episode_length, use_same_share_obs,
                 use_avail_acts, use_reward_normalization=False, device=None, device_tree=False):
        """device_tree=True keeps the sum/min trees in HBM (utils/device_per.py): sample() then returns the importance weights and
        the indices as device tensors, and update_priorities() takes device tensors, so a prioritized update has no host sync."""
        super(PrioritizedRecReplayBuffer, self).__init__(policy_info, policy_agents, buffer_size, episode_length,
                                                         use_same_share_obs, use_avail_acts, use_reward_normalization,
                                                         device=device)
        self.alpha = alpha
        self.device_tree = bool(device_tree)
        it_capacity = 1
        while it_capacity < buffer_size:
            it_capacity *= 2
        if self.device_tree:
            from .device_per import DevicePerTree
            dev = self.policy_buffers[next(iter(self.policy_info))].device
            self._dtrees = {p_id: DevicePerTree(it_capacity, alpha, dev) for p_id in self.policy_info.keys()}
        else:
            self._it_sums = {p_id: SumSegmentTree(it_capacity) for p_id in self.policy_info.keys()}
            self._it_mins = {p_id: MinSegmentTree(it_capacity) for p_id in self.policy_info.keys()}
        self.max_priorities = {p_id: 1.0 for p_id in self.policy_info.keys()}

    def insert(self, num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts=None):
        idx_range = super().insert(num_insert_episodes, obs, share_obs, acts, rewards, dones, dones_env, avail_acts)
        for p_id in self.policy_info.keys():      # A-3 fix: every new slot, not range(idx[0], idx[1])
            if self.device_tree:
                self._dtrees[p_id].set_to_max(idx_range)
                continue
            self._it_sums[p_id][idx_range] = self.max_priorities[p_id] ** self.alpha
            self._it_mins[p_id][idx_range] = self.max_priorities[p_id] ** self.alpha
        return idx_range

    def _sample_proportional(self, batch_size, p_id=None):
        total = self._it_sums[p_id].sum(0, len(self) - 1)
        mass = np.random.random(size=batch_size) * total
        return self._it_sums[p_id].find_prefixsum_idx(mass)

    def sample(self, batch_size, beta=0, p_id=None, shard=None):
        """rec_buffer.py:285-304. Data-parallel use: with `shard=(rank, world)` every rank draws the same `batch_size` global
        indices from its replica of the trees and gets back the episodes and importance weights of ITS contiguous share
        plus the GLOBAL index list; after the update, `dist.allgather_cat(new_priorities)` (the trainers return the LOCAL share's
        priorities; `have=trainer.gathered_priorities` skips the collective where the trainer gathered them itself) rebuilds the global priority
        vector so that `update_priorities(global_idxes, ...)` changes every replica identically."""
        assert len(self) > batch_size, "Cannot sample with no completed episodes in the buffer!"
        assert beta > 0
        if self.device_tree:    # same host RNG draw as _sample_proportional; tree walk and weights on the device
            inds, weights = self._dtrees[p_id].sample(np.random.random(size=batch_size), len(self), beta)
            if shard is not None:
                return self._gather(_shard(inds, *shard).contiguous()) + (_shard(weights, *shard).contiguous(), inds)
            return self._gather(inds) + (weights, inds)
        batch_inds = self._sample_proportional(batch_size, p_id)
        p_min = self._it_mins[p_id].min() / self._it_sums[p_id].sum()
        max_weight = (p_min * len(self)) ** (-beta)
        p_sample = self._it_sums[p_id][batch_inds] / self._it_sums[p_id].sum()
        weights = (p_sample * len(self)) ** (-beta) / max_weight
        if shard is not None:
            return self._gather(_shard(batch_inds, *shard)) + (_shard(weights, *shard), batch_inds)
        return self._gather(batch_inds) + (weights, batch_inds)

    def update_priorities(self, idxes, priorities, p_id=None):
        if self.device_tree:    # range checks of the host path would force a device sync; the kernels clamp nothing: callers pass
            self._dtrees[p_id].set(idxes, priorities)   # indices returned by sample()
            return
        priorities = np.asarray(priorities)
        idxes = np.asarray(idxes)
        assert len(idxes) == len(priorities)
        assert np.min(priorities) > 0
        assert np.min(idxes) >= 0
        assert np.max(idxes) < len(self)
        self._it_sums[p_id][idxes] = priorities ** self.alpha
        self._it_mins[p_id][idxes] = priorities ** self.alpha
        self.max_priorities[p_id] = max(self.max_priorities[p_id], np.max(priorities))
