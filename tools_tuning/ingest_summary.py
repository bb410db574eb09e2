"""Condense what tools_tuning/prof_ingest_all.sh collected under <dir>: the kernel statistics, the HBM traffic per dispatch (raw PMC
averages -> pmc_traffic_raw.json) and the SQ counters of the ingest's kernels.   usage: ingest_summary.py <dir>"""
import collections
import csv
import glob
import json
import sys

O = sys.argv[1]


def short(n):
    for k in ("ingest_planes_kernel", "cigar_runs_kernel", "cigar_walk_kernel", "ingest_init_kernel"):
        if k in n:
            rest = n[n.index(k) + len(k):]      # (the template arguments of the kernel itself, not of its parameters' types)
            return k + rest[:rest.index(">") + 1] if rest.startswith("<") else k
    return None


print("kernel statistics (rocprofv3 --kernel-trace --stats):")
for r in csv.DictReader(open(O + "/kernel_stats.csv")):
    if short(r["Name"]):
        print(f"   {short(r['Name']):50s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}")
traffic = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(O + f"/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        traffic.setdefault(k, {})[c + "_KB_avg"] = sum(v) / len(v)
        traffic[k]["dispatches"] = len(v)
print("HBM traffic per dispatch (KB as the counters give them; FETCH_SIZE counts 64-byte requests as 32 on gfx950 wide streams: see MI355X_MICROARCH.md):")
print(json.dumps(traffic, indent=1))
json.dump(traffic, open(O + "/pmc_traffic_raw.json", "w"), indent=1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {sum(v)/len(v):18.1f}  ({len(v)} dispatches)")
