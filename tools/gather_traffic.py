"""HBM traffic of the gather kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes: they do not fit one pass), merged into profiles/gather_traffic.json.

    python tools/gather_traffic.py <key workload:batch:episodes> <algorithmic_bytes> <fetch_counter_collection.csv> <write_counter_collection.csv>

FETCH_SIZE / WRITE_SIZE are KiB. On gfx950 FETCH_SIZE reports half the bytes of wide (16 B / lane) coalesced reads ->
doubled (the guide's correction); WRITE_SIZE is taken as is."""
import csv
import json
import os
import sys


def avg(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and "episode_copy_kernel<true" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


key, algo = sys.argv[1], int(float(sys.argv[2]))
f, nf = avg(sys.argv[3], "FETCH_SIZE")
w, nw = avg(sys.argv[4], "WRITE_SIZE")
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "gather_traffic.json")
d = json.load(open(path))
d["entries"][key] = {"algorithmic_bytes": algo, "fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1),
                     "traffic_bytes": int(2 * f * 1024 + w * 1024), "launches_averaged": min(nf, nw),
                     "note": "round 2 gather (episode-contiguous step path + LDS tiles), HBM-resident store"}
json.dump(d, open(path, "w"), indent=1)
print(key, d["entries"][key])
