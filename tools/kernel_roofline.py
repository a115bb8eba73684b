"""Per-kernel share of the f32 matrix roof for a QMIX-RNN bench workload: rocprofv3 kernel durations x analytic FLOP per launch.

    python tools/kernel_roofline.py <workload> <batch> <kernel_stats.csv> [<pmc_inst_table.txt>] [--note "..."]

`kernel_stats.csv` is the `rocprofv3 --kernel-trace --stats` summary of `bench.py --workload <workload> --batch <batch>` (Name, Calls,
TotalDurationNs, AverageNs, ...). The FLOP of a launch is the GEMM-shaped work the kernel performs per training step (the terms of
SURVEY.md section 8(d), split by kernel; LayerNorm / gates / elementwise excluded); `tflops` = FLOP / average duration, `frac` = that /
157.3 TFLOP/s (dense f32 MFMA = packed f32 vector peak of MI355X). With a `tools/pmc_table.py` table of a `--pmc SQ_INSTS_VALU_MFMA_MOPS_F32
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...` pass of the same command, `mfma_busy` = MFMA instructions x 32 cycles (v_mfma_f32_16x16x4_f32 issues
every 32 cycles per SIMD) / (duration x 2.4 GHz x 1 024 SIMDs) is added: the fraction of the launch during which the matrix pipes issue.
Writes / updates the entry "<workload>:<batch>" of profiles/kernel_roofline.json, which bench.py copies into `roofline_step.per_kernel`."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 157.3e12
SIMDS, CLK = 1024, 2.4e9


def flops(dims, B):
    N, A, D, S, T = dims.n_agents, dims.act_dim, dims.obs_dim, dims.state_dim, dims.episode_length
    R, R1, TB, NM = (T + 1) * N * B, T * N * B, T * B, N * 32
    trunk = 2 * R * (D * 64 + 64 * 64 + 64 * 192)
    mixer_mac = (S * 64 + 64 * NM) + (S * 64 + 64 * 32) + S * 32 + (S * 64 + 64) + (NM + 32)
    first = 2 * 2 * TB * (S * 224)                       # first hyper-layers of both nets
    f = {
        "trunk_fwd": trunk,                              # per launch (live and target are separate launches)
        "trunk_fwd_pair": 2 * trunk,                     # trunk_fwd4: both nets in one launch
        "trunk_bwd": 2 * R1 * (192 * 64 + 64 * 64),
        "gru_fwd": 2 * (2 * N * B) * (T + 1) * 192 * 64,  # live + target rows in one launch
        "gru_bwd": 2 * (N * B) * T * 192 * 64,
        "head_fwd": 2 * 2 * R * 64 * A,
        "head_bwd": 2 * R1 * 64 * A,
        "mixer_fwd": 2 * 2 * TB * mixer_mac,
        "mixer_wide_gemm": first,
        "mixer_fwd_stage2": 2 * 2 * TB * mixer_mac - first,
        "mixer_bwd": 2 * TB * (NM * 64 + 32 * 64),
        "wgrad": 2 * R1 * (64 * D + 64 * 64 + 192 * 64 + 128 * 64 + 64 * 64 + A * 64) + 2 * TB * (224 * S + NM * 64 + 32 * 64 + 64),
    }
    return f


def classify(name, have_wide):
    n = name
    if "mixer_wide_gemm" in n:
        return "mixer_wide_gemm"
    if "mixer_fwd" in n:
        return "mixer_fwd_stage2" if have_wide else "mixer_fwd"
    for key, pat in (("trunk_fwd_pair", "trunk_fwd4"), ("trunk_fwd", "trunk_fwd"), ("trunk_bwd", "trunk_bwd"), ("gru_fwd", "gru_fwd"), ("gru_bwd", "gru_bwd"),
                     ("head_fwd", "head_fwd"), ("head_bwd", "head_bwd"), ("mixer_bwd", "mixer_bwd"), ("wgrad", "wgrad_kernel")):
        if pat in n:
            return key
    return None


def short(name):
    return re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", "").replace("ope::", ""))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    note = sys.argv[sys.argv.index("--note") + 1] if "--note" in sys.argv else ""
    if "--note" in sys.argv:
        args = [a for a in args if a != note]
    workload, batch, stats = args[0], int(args[1]), args[2]
    pmc = args[3] if len(args) > 3 else None
    from offpolicy_amd.utils.synth import DIMS
    dims = DIMS[workload]
    fl = flops(dims, batch)
    rows = list(csv.DictReader(open(stats)))
    have_wide = any("mixer_wide_gemm" in r["Name"] for r in rows)
    mfma = {}
    if pmc:
        lines = open(pmc).read().splitlines()
        hdr = lines[0].split()
        # columns are right-aligned names truncated to 14 characters: find the MFMA instruction column
        col = next((i for i, h in enumerate(hdr) if h.endswith("SQ_INSTS_MFMA")), None)
        for ln in lines[1:]:
            m = re.match(r"(.{30}) (.*)", ln)
            if not m or col is None:
                continue
            vals = m.group(2).split()
            try:
                mfma[m.group(1).strip()] = float(vals[col - 1])
            except (ValueError, IndexError):
                pass
    steps = None
    out = {}
    total_ns = 0.0
    for r in rows:
        name = short(r["Name"])
        if any(k in name for k in ("at::native", "elementwise", "Memcpy", "fill")) and "ope" not in r["Name"]:
            continue
        calls, avg = int(r["Calls"]), float(r["AverageNs"])
        key = classify(name, have_wide)
        if "adam_kernel" in name:
            steps = calls
        ent = {"avg_us": round(avg / 1e3, 2), "calls": calls}
        if key:
            ent["flop"] = int(fl[key])
            ent["tflops"] = round(fl[key] / (avg * 1e-9) / 1e12, 2)
            ent["frac"] = round(fl[key] / (avg * 1e-9) / PEAK, 4)
        for pk, v in mfma.items():
            if name.startswith(pk[:28]) or pk.startswith(name[:28]):
                ent["mfma_insts"] = int(v)
                ent["mfma_busy"] = round(v * 32.0 / (avg * 1e-9 * CLK * SIMDS), 4)
        out[name[:60]] = ent
    # per-step totals (several launches of one kernel per step: trunk_fwd live / target are separate names; others once)
    if steps:
        for name, e in out.items():
            e["per_step"] = round(e["calls"] / float(steps), 2)
            total_ns += e["avg_us"] * 1e3 * e["per_step"]
    path = os.path.join(ROOT, "profiles", "kernel_roofline.json")
    d = json.load(open(path)) if os.path.exists(path) else {"peak_tflops": PEAK / 1e12, "entries": {}}
    d["entries"]["%s:%d" % (workload, batch)] = {"source": os.path.relpath(os.path.abspath(stats), ROOT), "pmc": pmc and os.path.relpath(os.path.abspath(pmc), ROOT),
                                                 "kernel_us_per_step": round(total_ns / 1e3, 1) if steps else None, "note": note, "kernels": out}
    json.dump(d, open(path, "w"), indent=1)
    for name, e in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1].get("per_step", 1)):
        print("%-58s %8.2f us x%5.2f  %s" % (name, e["avg_us"], e.get("per_step", 0), ("%6.1f TF/s = %5.1f %% of the f32 roof" % (e["tflops"], 100 * e["frac"])) if "frac" in e else "")
              + (("  mfma pipe busy %5.1f %%" % (100 * e["mfma_busy"])) if "mfma_busy" in e else ""))
    print("kernels per step: %.1f us" % (total_ns / 1e3))


if __name__ == "__main__":
    main()
