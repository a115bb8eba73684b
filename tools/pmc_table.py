"""Per-kernel averages of the counters in a rocprofv3 --pmc counter_collection.csv. Usage: pmc_table.py <csv> [filter]"""
import collections, csv, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("ope::", ""))[:30]
    d[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in d.values() for c in k})
flt = sys.argv[2] if len(sys.argv) > 2 else ""
print("%-30s " % "kernel" + " ".join("%14s" % c[-14:] for c in names))
for n, c in d.items():
    if flt and flt not in n:
        continue
    print("%-30s " % n + " ".join("%14.0f" % (sum(c[k]) / len(c[k])) if k in c else "%14s" % "-" for k in names))
