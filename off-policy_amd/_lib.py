"""ctypes binding of libope.so (the C-ABI in include/ope.h).

The library is the ONLY implementation of the update path: if it is missing or does not load, importing this
module raises -- there is no eager-PyTorch or CPU fallback.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede loading libope.so: torch's ROCm wheel bundles its own libamdhip64; loading
#                          ours first would put a second HIP runtime in the process and every launch on a torch stream fails.

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPE_LIB_PATH: load another build of the same library (the sanitizer build libope_asan.so in tests/test_sanitizer_host.py)
LIB_PATH = os.environ.get("OPE_LIB_PATH") or os.path.join(_HERE, "libope.so")

OPE_QMIX_NPARAM_AGENT = 22
OPE_QMIX_NPARAM_AGENT_2 = 26      # layer_N = 2
OPE_QMIX_NPARAM_AGENT_MLP = 16
OPE_QMIX_NPARAM_MIXER = 14
OPE_QMIX_NPARAM_MIXER_1 = 10      # hypernet_layers = 1
OPE_GRAD_TAIL = 4


class OpeError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [("n_agents", C.c_int32), ("act_dim", C.c_int32), ("obs_dim", C.c_int32), ("state_dim", C.c_int32),
                ("episode_length", C.c_int32), ("layer_N", C.c_int32), ("flags", C.c_int32)]


OPE_DIMS_NO_FEATURE_NORM = 1
OPE_DIMS_TANH = 2
OPE_DIMS_MASK_TARGET_MAX = 4


class Fields(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("obs", "share_obs", "acts", "rewards", "dones", "dones_env", "avail_acts", "valid_transition")]


class QmixCfg(C.Structure):
    _fields_ = [("dims", Dims), ("batch", C.c_int32), ("vdn", C.c_int32), ("use_double_q", C.c_int32),
                ("use_huber", C.c_int32), ("use_per", C.c_int32), ("gamma", C.c_float), ("huber_delta", C.c_float),
                ("per_nu", C.c_float), ("per_eps", C.c_float), ("mlp", C.c_int32), ("phase", C.c_int32),
                ("mixer_path", C.c_int32), ("time_chunks", C.c_int32), ("scan_family", C.c_int32), ("scan_waves", C.c_int32),
                ("debug", C.c_int32), ("trunk_path", C.c_int32), ("chain_path", C.c_int32), ("hypernet_layers", C.c_int32), ("wgrad_path", C.c_int32),
                ("live_rows", C.c_int32)]


class LiveTarget(C.Structure):
    _fields_ = [("plan", C.c_void_p), ("err_abs", C.c_void_p), ("loss_part", C.c_void_p), ("n_loss_part", C.c_int32),
                ("n_agents", C.c_int32), ("episode_length", C.c_int32), ("batch", C.c_int32), ("copy_live_only", C.c_int32)]


class GatherTune(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("floats_per_block", "xcd_run", "unroll", "nontemporal", "small_tiles", "tile_floats")]


class ObsRef(C.Structure):
    """ope_obs_ref (include/ope.h): a batch's observation rows left in the replay store."""
    _fields_ = [("store_obs", C.c_void_p), ("inds", C.c_void_p), ("capacity", C.c_int32), ("reserved", C.c_int32)]


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("max_grad_norm", C.c_float), ("weight_decay", C.c_float), ("tau", C.c_float),
                ("do_polyak", C.c_int32), ("step", C.c_int32), ("qtot_denominator", C.c_float), ("tail_offset", C.c_int32),
                ("step_counter", C.c_void_p), ("skip_begin", C.c_int32), ("skip_end", C.c_int32),
                ("sumsq_partials", C.c_void_p), ("n_sumsq_partials", C.c_int32)]


class DdpgOpt(C.Structure):
    """ope_ddpg_opt (include/ope.h): the optimiser step carried by a fused MADDPG / MATD3 update launch."""
    _fields_ = [("adam", AdamCfg), ("n", C.c_int64), ("theta_tgt", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
                ("stats_out", C.c_void_p)]


class DdpgCfg(C.Structure):
    _fields_ = [("dims", Dims), ("batch", C.c_int32), ("num_q", C.c_int32), ("target_gumbel", C.c_int32),
                ("use_huber", C.c_int32), ("use_per", C.c_int32), ("gamma", C.c_float), ("huber_delta", C.c_float),
                ("per_eps", C.c_float), ("noise_seed", C.c_uint64), ("noise_counter", C.c_void_p),
                ("n_total_agents", C.c_int32), ("agent_offset", C.c_int32), ("joint_next_acts", C.c_void_p),
                ("continuous", C.c_int32), ("n_act_heads", C.c_int32), ("act_head_dims", C.c_int32 * 6),
                ("joint_act_dim", C.c_int32), ("joint_act_col", C.c_int32), ("joint_acts", C.c_void_p)]


class RddpgCfg(C.Structure):
    _fields_ = [("dims", Dims), ("batch", C.c_int32), ("num_q", C.c_int32), ("target_gumbel", C.c_int32),
                ("use_huber", C.c_int32), ("use_per", C.c_int32), ("gamma", C.c_float), ("huber_delta", C.c_float),
                ("n_total_agents", C.c_int32), ("agent_offset", C.c_int32), ("joint_next_acts", C.c_void_p), ("actor_row_weight", C.c_void_p),
                ("continuous", C.c_int32), ("n_act_heads", C.c_int32), ("act_head_dims", C.c_int32 * 6),
                ("joint_act_dim", C.c_int32), ("joint_act_col", C.c_int32), ("joint_acts", C.c_void_p)]


class AllreduceCtx(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("max_floats", C.c_int64), ("peer", C.c_void_p * 16), ("timeout_ms", C.c_int32)]


class MlpBatch(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("obs", "share_obs", "acts", "rewards", "next_obs", "next_share_obs", "dones_env",
                                          "valid_transition", "avail_acts", "next_avail_acts")]


ABI_MIRRORS = {"ope_live_target": LiveTarget, "ope_dims": Dims, "ope_fields": Fields, "ope_qmix_cfg": QmixCfg, "ope_gather_tune": GatherTune, "ope_obs_ref": ObsRef,
               "ope_adam_cfg": AdamCfg, "ope_ddpg_opt": DdpgOpt, "ope_ddpg_cfg": DdpgCfg, "ope_rddpg_cfg": RddpgCfg,
               "ope_allreduce_ctx": AllreduceCtx, "ope_mlp_batch": MlpBatch}


def _load():
    if not os.path.exists(LIB_PATH):
        raise OpeError("libope.so not found at %s -- build it with `python -m offpolicy_amd.build` "
                       "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    p = C.c_void_p
    i32, i64 = C.c_int32, C.c_int64
    sig = {
        "ope_version": (C.c_int, []),
        "ope_abi_sizeof": (C.c_int64, [C.c_char_p]),
        "ope_strerror": (C.c_char_p, [C.c_int]),
        "ope_set_debug": (None, [C.c_int]),
        "ope_set_scan_kernel": (None, [C.c_int, C.c_int]),
        "ope_set_w2_fin": (None, [C.c_int]),
        "ope_last_launches": (C.c_int, [C.c_char_p, i32]),
        "ope_kernel_profile": (C.c_int, [i32, i32]),
        "ope_kernel_profile_read": (C.c_int, [C.c_char_p, i32]),
        "ope_episode_bytes": (i64, [C.POINTER(Dims)]),
        "ope_store_insert": (C.c_int, [C.POINTER(Dims), i32, C.POINTER(Fields), C.POINTER(Fields), p, i32, p, p]),
        "ope_store_gather": (C.c_int, [C.POINTER(Dims), i32, C.POINTER(Fields), p, i32, C.POINTER(Fields), p, p]),
        "ope_store_gather_host_inds": (C.c_int, [C.POINTER(Dims), i32, C.POINTER(Fields), p, i32, C.POINTER(Fields), p]),
        "ope_store_gather_sampled": (C.c_int, [C.POINTER(Dims), i32, i32, p, C.POINTER(Fields), C.c_uint64, p, i32, C.POINTER(Fields), p, p]),
        "ope_store_gather_tuned": (C.c_int, [C.POINTER(Dims), i32, C.POINTER(Fields), p, p, i32, C.POINTER(Fields), p, C.POINTER(GatherTune), p]),
        "ope_store_gather_ref": (C.c_int, [C.POINTER(Dims), i32, C.POINTER(Fields), p, p, i32, C.POINTER(Fields), p, p, C.POINTER(GatherTune), p]),
        "ope_qmix_obs_ref_ok": (C.c_int, [C.POINTER(QmixCfg)]),
        "ope_qmix_live_rows_ok": (C.c_int, [C.POINTER(QmixCfg)]),
        "ope_qmix_live_plan": (C.c_int, [C.POINTER(QmixCfg), p, p, i64, p]),
        "ope_qmix_live_target": (C.c_int, [C.POINTER(QmixCfg), p, i64, i32, C.POINTER(LiveTarget)]),
        "ope_store_live_plan": (C.c_int, [i32, i32, p, p, p, C.POINTER(LiveTarget), p]),
        "ope_store_gather_attach_live": (C.c_int, [C.POINTER(LiveTarget)]),
        "ope_qmix_signal_event": (C.c_int, [p, i32]),
        "ope_qmix_loss_and_grad_ref": (C.c_int, [C.POINTER(QmixCfg), C.POINTER(Fields), C.POINTER(ObsRef), p, p, p, p, i64, p, p, p]),
        "ope_store_gather_profile": (C.c_int, [i32]),
        "ope_store_gather_profile_read": (C.c_int, [p, i32]),
        "ope_set_gather_params": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "ope_per_tree_bytes": (i64, [i32]),
        "ope_per_tree_init": (C.c_int, [p, i32, p]),
        "ope_per_tree_set": (C.c_int, [p, i32, p, p, C.c_double, i32, p]),
        "ope_per_tree_sample": (C.c_int, [p, i32, i32, p, C.c_double, i32, p, p, p]),
        "ope_per_tree_sample_dev": (C.c_int, [p, i32, p, p, p, i32, p, p, p]),
        "ope_reward_stats_scratch_bytes": (i64, []),
        "ope_store_reward_stats": (C.c_int, [C.POINTER(Dims), i32, p, p, p, p, p]),
        "ope_reward_normalize": (C.c_int, [p, i64, p, p]),
        "ope_qmix_param_layout": (i64, [C.POINTER(QmixCfg), C.POINTER(i64), C.POINTER(i64)]),
        "ope_qmix_workspace_bytes": (i64, [C.POINTER(QmixCfg)]),
        "ope_qmix_workspace_init": (C.c_int, [C.POINTER(QmixCfg), p, i64, p]),
        "ope_qmix_workspace_find": (i64, [C.POINTER(QmixCfg), C.c_char_p, C.POINTER(i64)]),
        "ope_qmix_loss_and_grad": (C.c_int, [C.POINTER(QmixCfg), C.POINTER(Fields), p, p, p, p, i64, p, p, p]),
        "ope_agent_forward_workspace_bytes": (i64, [C.POINTER(Dims), i32, i32]),
        "ope_agent_forward": (C.c_int, [C.POINTER(Dims), i32, i32, p, p, p, p, i64, p, p, p]),
        "ope_agent_forward_mlp_workspace_bytes": (i64, [C.POINTER(Dims), i32]),
        "ope_agent_forward_mlp": (C.c_int, [C.POINTER(Dims), i32, p, p, p, i64, p, p]),
        "ope_ddpg_param_layout": (i64, [C.POINTER(DdpgCfg), i32, C.POINTER(i64), C.POINTER(i64)]),
        "ope_ddpg_workspace_bytes": (i64, [C.POINTER(DdpgCfg)]),
        "ope_ddpg_workspace_init": (C.c_int, [C.POINTER(DdpgCfg), p, i64, p]),
        "ope_ddpg_workspace_find": (i64, [C.POINTER(DdpgCfg), C.c_char_p, C.POINTER(i64)]),
        "ope_ddpg_critic_loss_and_grad": (C.c_int, [C.POINTER(DdpgCfg), C.POINTER(MlpBatch), p, p, p, p, p, p, i64, p, p, p]),
        "ope_ddpg_update_ok": (C.c_int, [C.POINTER(DdpgCfg)]),
        "ope_ddpg_critic_update": (C.c_int, [C.POINTER(DdpgCfg), C.POINTER(MlpBatch), p, p, p, p, p, p, i64, p, p, C.POINTER(DdpgOpt), p]),
        "ope_ddpg_actor_update": (C.c_int, [C.POINTER(DdpgCfg), C.POINTER(MlpBatch), p, p, p, p, i64, p, C.POINTER(DdpgOpt), p]),
        "ope_ddpg_target_actions": (C.c_int, [C.POINTER(DdpgCfg), C.POINTER(MlpBatch), p, p, p, i64, p, p]),
        "ope_ddpg_actor_loss_and_grad": (C.c_int, [C.POINTER(DdpgCfg), C.POINTER(MlpBatch), p, p, p, p, i64, p, p]),
        "ope_rddpg_param_layout": (i64, [C.POINTER(RddpgCfg), i32, C.POINTER(i64), C.POINTER(i64)]),
        "ope_rddpg_workspace_bytes": (i64, [C.POINTER(RddpgCfg)]),
        "ope_rddpg_workspace_init": (C.c_int, [C.POINTER(RddpgCfg), p, i64, p]),
        "ope_rddpg_workspace_find": (i64, [C.POINTER(RddpgCfg), C.c_char_p, C.POINTER(i64)]),
        "ope_rddpg_target_actions": (C.c_int, [C.POINTER(RddpgCfg), C.POINTER(Fields), p, p, p, i64, p, p]),
        "ope_rddpg_critic_loss_and_grad": (C.c_int, [C.POINTER(RddpgCfg), C.POINTER(Fields), p, p, p, p, p, p, i64, p, p, p]),
        "ope_rddpg_actor_loss_and_grad": (C.c_int, [C.POINTER(RddpgCfg), C.POINTER(Fields), p, p, p, p, i64, p, p]),
        "ope_adam_scratch_floats": (i64, [i64]),
        "ope_adam_step": (C.c_int, [C.POINTER(AdamCfg), i64, p, p, p, p, p, p, p, p]),
        "ope_polyak": (C.c_int, [i64, p, p, C.c_float, p]),
        "ope_allreduce_buffer_bytes": (i64, [i64, i32]),
        "ope_allreduce_alloc": (C.c_int, [i64, C.POINTER(p)]),
        "ope_allreduce_free": (C.c_int, [p]),
        "ope_allreduce_ipc_export": (C.c_int, [p, p]),
        "ope_allreduce_ipc_import": (C.c_int, [p, C.POINTER(p)]),
        "ope_allreduce_ipc_close": (C.c_int, [p]),
        "ope_allreduce_enable_peer": (C.c_int, [i32]),
        "ope_allreduce_flat": (C.c_int, [C.POINTER(AllreduceCtx), C.c_uint32, p, i64, p, p]),
        "ope_allreduce_flat_dev": (C.c_int, [C.POINTER(AllreduceCtx), p, p, i64, p, p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = a symbol declared in include/ope.h is missing
        fn.restype = res
        fn.argtypes = args
    # the mirror declarations above against the structs the library was compiled with: a drifted field would be read as garbage silently
    for cname, cls in ABI_MIRRORS.items():
        want = lib.ope_abi_sizeof(cname.encode())
        if want != C.sizeof(cls):
            raise OpeError("%s: sizeof is %d in libope.so but %d in offpolicy_amd._lib.%s -- rebuild the library or fix the mirror"
                           % (cname, want, C.sizeof(cls), cls.__name__))
    return lib, sorted(sig)


lib, EXPORTS = _load()


def last_launches():
    """Kernel variants the last ope_qmix_loss_and_grad call of this thread launched, in launch order (ope_last_launches)."""
    buf = C.create_string_buffer(2048)
    lib.ope_last_launches(buf, 2048)
    return [x for x in buf.value.decode().split(";") if x]


def kernel_profile(enable, max_launches=8192):
    check(lib.ope_kernel_profile(1 if enable else 0, int(max_launches)), "ope_kernel_profile")


def kernel_profile_read(with_rows=False):
    """[(demangled kernel name, calls, total_ms, min_ms, max_ms, flop, bytes)] of the launches since kernel_profile(True), in order of first
    launch; flop / bytes = the algorithmic work of those launches as their launchers state it (0 where none is stated). with_rows: an 8th
    entry, the live-row count that scales the stated work of a launch on live rows (0 none; ope.h, ope_kernel_profile_read)."""
    buf = C.create_string_buffer(1 << 18)
    n = lib.ope_kernel_profile_read(buf, 1 << 18)
    if n < 0:
        check(n, "ope_kernel_profile_read")
    out = []
    for ln in buf.value.decode().splitlines():
        name, calls, tot, mn, mx, flop, nbytes, rows = ln.rsplit("\t", 7)
        rec = (name, int(calls), float(tot), float(mn), float(mx), float(flop), float(nbytes))
        out.append(rec + (int(rows),) if with_rows else rec)
    return out


def check(rc, what=""):
    if rc != 0:
        raise OpeError("%s failed: %s (%d)" % (what or "ope call", lib.ope_strerror(int(rc)).decode(), rc))


_raw_stream = None


def current_stream():
    """hipStream_t of torch's current stream (of the current device) as an integer handle. Goes through torch's raw-stream
    accessor when there is one: torch.cuda.current_stream() builds a Stream object behind four Python-level device lookups
    (~9 us), and a training step asks five times."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        fn = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        _raw_stream = fn if fn is not None else False
    if _raw_stream:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor, or NULL for None."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "ope kernels need contiguous tensors"
    return C.c_void_p(t.data_ptr())
