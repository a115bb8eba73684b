"""Run by tests/test_sanitizer_host.py in a subprocess with the AddressSanitizer runtime preloaded and OPE_LIB_PATH = libope_asan.so:
every C-ABI entry point that finishes on the host (parameter layouts, workspace planning / lookup, size queries, argument checks that
return before the first launch) over a spread of shapes -- tiny, odd, the BASELINE configs, the wide-state ones, multi-policy phases --,
so that the host-side table building, Workspace bookkeeping and plan arithmetic run under ASan + UBSan. No GPU is touched."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offpolicy_amd import _lib  # noqa: E402

L = _lib.lib
assert os.path.basename(_lib.LIB_PATH) == "libope_asan.so", _lib.LIB_PATH
assert L.ope_version() >= 1
for code in (0, -1, -2, -3, -4, -99):
    assert L.ope_strerror(code)
n_calls = 0
off, siz = (C.c_int64 * 64)(), (C.c_int64 * 64)()
shapes = [(2, 5, 12, 10, 6), (3, 7, 18, 54, 5), (3, 9, 64, 48, 60), (8, 14, 252, 216, 150), (10, 18, 370, 322, 180), (8, 14, 252, 2232, 150),
          (3, 9, 64, 240, 12), (3, 7, 18, 83, 5), (1, 1, 1, 1, 1), (64, 200, 512, 4022, 3)]
names = [b"agent_q", b"agent_nq", b"d_agent_q", b"gsq_part", b"q_all", b"dbg", b"h", b"qtot", b"err_abs", b"mix_slab", b"no_such_region"]
for (n, a, d, s, t) in shapes:
    dims = _lib.Dims(n, a, d, s, t)
    assert L.ope_episode_bytes(C.byref(dims)) > 0
    for batch in (1, 4, 32):
        for vdn in (0, 1):
            for mlp in (0, 1):
                for phase in (0, 1, 2, 3):
                    for path in (0, 1, 2, 3):
                        for chunks in (0, 3):
                            cfg = _lib.QmixCfg()
                            cfg.dims = _lib.Dims(n, a, d, s, 1 if mlp else t)
                            cfg.batch, cfg.vdn, cfg.mlp, cfg.phase, cfg.mixer_path, cfg.time_chunks = batch, vdn, mlp, phase, path, chunks
                            cfg.use_double_q, cfg.gamma = 1, 0.99
                            total = L.ope_qmix_param_layout(C.byref(cfg), off, siz)
                            need = L.ope_qmix_workspace_bytes(C.byref(cfg))
                            assert total > 0 and need > 0, (n, a, d, s, t, batch, vdn, mlp, phase, path)
                            for nm in names:
                                cnt = C.c_int64(0)
                                o = L.ope_qmix_workspace_find(C.byref(cfg), nm, C.byref(cnt))
                                assert o < need and (o < 0 or o + 4 * cnt.value <= need), nm
                            # argument checks that return before any launch
                            assert L.ope_qmix_loss_and_grad(C.byref(cfg), None, None, None, None, None, 0, None, None, None) == -1
                            assert L.ope_qmix_workspace_init(C.byref(cfg), None, 0, None) == -1
                            n_calls += 4 + len(names)
    for which in (0, 1):
        for num_q in (1, 2):
            dc = _lib.DdpgCfg()
            dc.dims, dc.batch, dc.num_q = _lib.Dims(n, a, min(d, 128), s, 1), 16, num_q
            rc = _lib.RddpgCfg()
            rc.dims, rc.batch, rc.num_q = dims, 4, num_q
            if L.ope_ddpg_param_layout(C.byref(dc), which, off, siz) > 0:
                assert L.ope_ddpg_workspace_bytes(C.byref(dc)) > 0
                cnt = C.c_int64(0)
                L.ope_ddpg_workspace_find(C.byref(dc), b"gsq_critic", C.byref(cnt))
            if L.ope_rddpg_param_layout(C.byref(rc), which, off, siz) > 0:
                assert L.ope_rddpg_workspace_bytes(C.byref(rc)) > 0
                cnt = C.c_int64(0)
                L.ope_rddpg_workspace_find(C.byref(rc), b"q", C.byref(cnt))
            n_calls += 6
# rejected shapes return error codes, never touch memory
bad = _lib.QmixCfg()
bad.dims, bad.batch = _lib.Dims(3, 9, 1000, 48, 60), 32
assert L.ope_qmix_param_layout(C.byref(bad), off, siz) == -1 and L.ope_qmix_workspace_bytes(C.byref(bad)) == -1
bad.dims, bad.mixer_path = _lib.Dims(3, 9, 64, 48, 60), 7
assert L.ope_qmix_workspace_bytes(C.byref(bad)) == -1
bad.mixer_path, bad.trunk_path = 0, 5
assert L.ope_qmix_workspace_bytes(C.byref(bad)) == -1
for cap in (1, 2, 64, 8192):
    assert L.ope_per_tree_bytes(cap) > 0
assert L.ope_per_tree_bytes(3) < 0 or L.ope_per_tree_bytes(3) > 0
assert L.ope_reward_stats_scratch_bytes() > 0
for nn in (1, 5, 118795, 620000):
    assert L.ope_adam_scratch_floats(nn) > 0
for world in (1, 2, 8, 16):
    assert L.ope_allreduce_buffer_bytes(1 << 18, world) > 0
assert L.ope_allreduce_buffer_bytes(1000, 2) == -1 and L.ope_allreduce_buffer_bytes(1 << 18, 17) == -1
assert L.ope_adam_step(None, 4, None, None, None, None, None, None, None, None) == -1
assert L.ope_polyak(0, None, None, 0.5, None) == -1
d0 = _lib.Dims(2, 5, 12, 10, 6)
assert L.ope_store_gather(C.byref(d0), 4, None, None, 2, None, None, None) == -1
import numpy as np  # noqa: E402
inds = np.array([0, 9], dtype=np.int64)          # 9 >= capacity 4: rejected on the host like numpy's IndexError
f = _lib.Fields()
assert L.ope_store_gather_host_inds(C.byref(d0), 4, C.byref(f), inds.ctypes.data_as(C.c_void_p), 2, C.byref(f), None) == -1
ctx = _lib.AllreduceCtx()
assert L.ope_allreduce_flat(C.byref(ctx), 0, None, 4, None, None) == -1
assert L.ope_allreduce_flat_dev(C.byref(ctx), None, None, 4, None, None) == -1          # no epoch state
assert L.ope_allreduce_flat_dev(None, None, None, 4, None, None) == -1
assert L.ope_per_tree_sample_dev(None, 64, None, None, None, 4, None, None, None) == -1
assert L.ope_per_tree_sample(None, 64, 8, None, 0.5, 4, None, None, None) == -1
assert L.ope_per_tree_sample(None, 3, 8, None, 0.5, 4, None, None, None) == -1            # capacity not a power of two
L.ope_set_debug(0)
L.ope_set_scan_kernel(4, 2)
L.ope_set_scan_kernel(0, 0)
L.ope_set_gather_params(0, 0, 0, -1, -1, 0)
print("SANITIZER_DRIVER_OK", n_calls)
