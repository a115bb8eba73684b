"""Per-kernel summary (calls, total, avg/min/max, share) out of a rocprofv3 results .db (rocpd sqlite output of
`rocprofv3 --kernel-trace --stats`). Usage: python tools/rocprof_db_stats.py <results.db> [out.csv]"""
import csv
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows))
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")]
    for r in rows:
        out.append((r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3)))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f).writerows(out)
    print("total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    for r in out[1:40]:
        print("%-60s calls=%5d total_ms=%9.3f avg_us=%9.2f min=%8.2f max=%8.2f pct=%5.2f" % (
            re.sub(r"\(.*", "", r[0])[:60], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6]))


if __name__ == "__main__":
    main()
