"""Timeline of the last few steps from a rocprofv3 kernel trace: start offset, duration and gap to the previous kernel.
usage: python tools_tuning/trace_gaps.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-46s start %8.2f us  dur %7.2f us  gap %6.2f us  grid %s wg %s" % (r["Kernel_Name"][:46], (s - t0) / 1e3, (e - s) / 1e3, gap,
          r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    prev_end = e
