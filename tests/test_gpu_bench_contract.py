"""-m gpu: bench.py's one-line JSON contract (what the round driver parses), on tiny settings."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len([ln for ln in lines if ln.lstrip().startswith("{")]) == 1, "exactly ONE JSON line on stdout"
    return json.loads(lines[-1])


def check_common(out, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == steps and out["warmup"] == warmup
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["higher_is_better"] is True
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) <= 0.02 * out["value"]      # steps/s of the whole job = 1 / step time at N = 1
    assert out["scaling"] in ("weak", "strong") and out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert isinstance(out["config"].get("workload"), str) and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0 and rf["achieved"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf


def test_default_workload_line_has_roofline_and_cpu_baseline():
    out = run_bench("--workload", "3m", "--episodes", "64", "--steps", "6", "--warmup", "2", "--cpu-seconds", "0.5")
    check_common(out, 6, 2)
    assert "QMIX" in out["metric"] and out["unit"] == "training steps/sec"
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and isinstance(cb["sample"], str) and cb["unit"] == out["unit"]
    assert out["value"] > 10 * cb["value"]                                           # the north-star bar, even on the tiny map
    tm = out["timing"]                                                               # K steps timed `--repeats` times, median reported
    assert tm["windows"] == 5 and tm["steps_per_window"] == 6 and len(tm["ms_per_step_windows"]) == 5
    assert tm["ms_per_step_min"] <= out["ms_per_step"] <= tm["ms_per_step_max"]
    rs = out["roofline_step"]                                                        # the whole step against the f32 matrix peak
    assert rs["bound"] == "mfma" and rs["peak"] == 157.3 and rs["flop_per_step"] == 4 * 2 * (61 * 3 * 32 * (64 * 64 + 64 * 64 + 6 * 64 * 64 + 64 * 9)
                                                                                      + 60 * 32 * ((48 * 64 + 64 * 96) + (48 * 64 + 64 * 32) + 48 * 32 + (48 * 64 + 64) + (96 + 32)))
    assert abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-3 and abs(rs["achieved"] - rs["flop_per_step"] / out["ms_per_step"] / 1e9) < 0.02 * rs["achieved"]
    # per-kernel table measured IN THIS RUN (hipExtLaunchKernel events on every libope launch + the launchers' stated work): the kernels of
    # the step are there, their durations add up to about the step, every GEMM-shaped one has a fraction of the matrix roof
    pk = rs["per_kernel"]
    names = [k["kernel"] for k in pk["kernels"]]
    for want in ("trunk_fwd", "gru_fwd4", "gru_bwd4", "episode_copy_kernel", "adam_kernel"):
        assert any(want in n for n in names), (want, names)
    # the weight gradients: the register-blocked launch + its slab sum (with the finalize step folded in, or followed by it), or the tile-per-wave one + split_reduce
    assert all(any(w in n for n in names) for w in ("wgrad2_kernel", "w2_fin_kernel")) or \
        all(any(w in n for n in names) for w in ("wgrad2_kernel", "w2_reduce_kernel")) or \
        all(any(w in n for n in names) for w in ("wgrad_kernel", "split_reduce_kernel")), names
    # the (t, b)-row chain: the fused pair, or the four separate launches
    assert all(any(w in n for n in names) for w in ("qchain_kernel", "mixer_hyp_kernel")) or \
        all(any(w in n for n in names) for w in ("mixer_fwd", "mixer_bwd4", "head_fwd_mfma", "head_bwd_rows")), names
    assert 0.5 * 1e3 * out["ms_per_step"] <= pk["kernel_us_per_step"] <= 1.3 * 1e3 * out["ms_per_step"]
    for k in pk["kernels"]:
        assert k["avg_us"] > 0 and k["launches_per_step"] > 0
        if k.get("bound") == "mfma":
            assert 0 < k["frac"] <= 1.0 and abs(k["tflops"] - k["flop_per_launch"] / k["avg_us"] / 1e6) <= 0.02 * k["tflops"] + 0.01
        if k.get("bound") == "hbm":
            assert 0 < k["frac"] <= 1.0
    # the launchers' own FLOP count agrees with SURVEY's closed form to ~10 % (SURVEY counts 2x the forward for the backward, which includes an input adjoint of fc1 that no gradient needs)
    assert abs(pk["launcher_stated_flop_per_step"] - rs["flop_per_step"]) <= 0.15 * rs["flop_per_step"]


def test_gall_workload_line():
    """The reference's own QMIX-SMAC launch configuration (scripts/train_smac_qmix.sh): wide state, gain 1, hard target updates."""
    out = run_bench("--workload", "3m_gall", "--episodes", "64", "--steps", "6", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline")
    check_common(out, 6, 2)
    assert "S=240" in out["config"]["workload"] and "hard target update" in out["config"]["workload"]


def test_maddpg_workload_line_graph_and_eager():
    for extra in ((), ("--no-graph",)):
        out = run_bench("--workload", "maddpg_spread", "--steps", "8", "--warmup", "4", "--no-cpu-baseline", *extra)
        check_common(out, 8, 4)
        assert "MADDPG" in out["metric"] and "cpu_baseline" not in out
        rs = out["roofline_step"]                       # BASELINE config 3 carries its whole-step roofline and per-kernel table too
        assert rs is not None and rs["bound"] == "mfma" and rs["flop_per_step"] > 0 and 0 < rs["frac"] < 1
        names = [k["kernel"] for k in rs["per_kernel"]["kernels"]]
        assert any("ddpg_critic_tile_kernel" in n for n in names) and any("ddpg_actor_tile_kernel" in n for n in names), names


@pytest.mark.parametrize("workload,batch", [("maddpg_spread", 256), ("rmatd3_3m", 128)])
def test_two_rank_self_spawn_other_workloads(workload, batch):
    """The same two-rank self-spawn for the MADDPG-MLP and the recurrent MATD3 + PER legs (replicated sum / min trees, all-gathered
    priorities)."""
    env = dict(os.environ, OPE_BENCH_SELFTEST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    # (round 6: `value` is the STRONG leg -- BASELINE's fixed global batch sharded over the ranks --, the weak leg beside it)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == batch and out["config"]["batch_per_gpu"] == batch // 2 and out["config"]["allreduce"]
    assert out["weak_scaling"]["global_batch"] == 2 * batch and out["weak_scaling"]["batch_per_gpu"] == batch and out["weak_scaling"]["value"] > 0


def test_two_rank_self_spawn_line_on_one_gpu():
    """`python bench.py --gpus 2` without a launcher starts the two ranks itself. OPE_BENCH_SELFTEST=1 puts both on cuda:0 over gloo, so
    the N > 1 branches (rendezvous, index sharding, the gradient all-reduce between loss_and_grad and the optimizer, max-over-ranks timing,
    strong leg = `value` (ONE batch of 32 sharded over the ranks: BASELINE.json's configuration), weak leg as a second field) run on a 1-GPU box."""
    env = dict(os.environ, OPE_BENCH_SELFTEST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "3m", "--episodes", "64", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["value"] > 0
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 32 and out["config"]["batch_per_gpu"] == 16
    assert out["config"]["parallelism"] == "dp2" and out["config"]["allreduce"] and "sharded over 2 GPUs" in out["metric"]
    weak = out["weak_scaling"]
    assert weak["global_batch"] == 64 and weak["batch_per_gpu"] == 32 and weak["value"] > 0
    assert "cpu_baseline" not in out and out["roofline"]["achieved"] > 0


def test_two_rank_dry_run_reports_the_backend_per_rank():
    """`bench.py --gpus 2 --dry-run` (what to run first on a new multi-GPU node): rendezvous, all-reduce backend per rank, a verified and timed
    475 KB all-reduce, one JSON line -- and nothing else is built."""
    env = dict(os.environ, OPE_BENCH_SELFTEST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and len(out["ranks"]) == 2 and out["all_ok"] is True
    for q, rk in enumerate(out["ranks"]):
        assert rk["rank"] == q and rk["allreduce"] and rk["matches_torch_distributed"] and rk["allreduce_us"] > 0
    assert r.stderr.count("[bench dry-run]") == 2
