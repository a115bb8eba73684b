"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace CSV (start/end relative to the step start,
duration, stream). Usage: python tools/timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, re, sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("ope::", "")
    return re.sub(r"\(.*", "", n)[:34]


rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Stream_Id"]) for r in rows)
idx = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = idx[-back - 1], idx[-back]
t0 = ks[a + 1][0]
busy = sum(k[1] - k[0] for k in ks[a + 1:b + 1])
print("step span %.1f us, %d kernels, sum of kernel durations %.1f us" % ((ks[b][1] - t0) / 1e3, b - a, busy / 1e3))
prev_end = t0
for k in ks[a + 1:b + 1]:
    print("  start %8.1f  end %8.1f  dur %7.1f  gap_after_prev_start %6.1f  s=%s %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3,
                                                                              (k[0] - prev_end) / 1e3, k[3], k[2]))
    prev_end = max(prev_end, k[1])
