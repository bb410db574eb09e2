"""The timed region of `bench.py --steps 20 --warmup 5` in a rocprofv3 kernel trace: the last partial (4-window) group launch
and the two 8-window launches in front of it.  usage: python tools_tuning/trace_timed_region.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
z4 = [i for i, r in enumerate(rows) if "pileup_group" in r["Kernel_Name"] and r.get("Grid_Size_Z") == "4"]
last = z4[-1]
pg = [i for i, r in enumerate(rows[:last]) if "pileup_group" in r["Kernel_Name"]]
start = pg[-2]
t0 = int(rows[start]["Start_Timestamp"])
n_done = 0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-34s q%-2s z%-2s start %7.1f end %7.1f dur %6.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:34],
          r.get("Queue_Id", "?"), r.get("Grid_Size_Z", "?"), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
    if "done_group" in r["Kernel_Name"]:
        n_done += 1
        if n_done == 3:
            break
