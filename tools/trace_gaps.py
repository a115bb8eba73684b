"""Per-kernel duration and the idle gap in front of it, from a `rocprofv3 --kernel-trace --output-format csv` trace:
    python tools/trace_gaps.py <dir-or-csv> [--last N]
Averages over the steady-state tail of the run (the last N dispatches), grouped by position in the repeating step."""
import csv, glob, os, sys
from collections import OrderedDict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("ope::", "").replace("void ", "")
    return name.split("(")[0][:70]


def main():
    path = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 4000
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[-last:]
    agg = OrderedDict()
    prev_end = None
    for s, e, n in rows:
        gap = (s - prev_end) if prev_end is not None else 0
        prev_end = e
        a = agg.setdefault(n, [0, 0.0, 0.0])
        a[0] += 1; a[1] += e - s; a[2] += gap
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print("window %.1f us, busy %.1f us (%.1f%%), %d dispatches" % (span / 1e3, busy / 1e3, 100.0 * busy / span, len(rows)))
    print("%-72s %7s %9s %9s" % ("kernel", "calls", "avg us", "gap us"))
    for n, (c, d, g) in agg.items():
        print("%-72s %7d %9.2f %9.2f" % (n, c, d / c / 1e3, g / c / 1e3))


if __name__ == "__main__":
    main()
