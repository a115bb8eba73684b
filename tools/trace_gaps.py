"""Gaps between consecutive kernel dispatches of a rocprofv3 --kernel-trace CSV: mean idle time by (previous kernel -> next kernel).
    python tools/trace_gaps.py <kernel_trace.csv> [first_fraction last_fraction]
"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(lo * len(rows)):int(hi * len(rows))]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|ope::|void ", "", n)
    return n.split("(")[0][:48]
gaps = collections.defaultdict(list)
durs = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
for r in rows:
    durs[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-50s %-50s %6s %9s" % ("after", "before", "n", "gap us"))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]) * (len(kv[1]) > 5)):
    if len(v) > 5:
        print("%-50s %-50s %6d %9.2f" % (k[0], k[1], len(v), sum(v) / len(v)))
print()
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    print("%-60s n %6d  avg %8.2f us" % (k, len(v), sum(v) / len(v)))
