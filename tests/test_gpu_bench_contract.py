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


def test_maddpg_workload_line_graph_and_eager():
    for extra in ((), ("--no-graph",)):
        out = run_bench("--workload", "maddpg_spread", "--steps", "8", "--warmup", "4", "--no-cpu-baseline", *extra)
        check_common(out, 8, 4)
        assert "MADDPG" in out["metric"] and "cpu_baseline" not in out
