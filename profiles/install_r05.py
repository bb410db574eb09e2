"""Copy what profiles/collect_r05.sh left under gpurun_out/r05 into profiles/r05_* (the files the README table names) and
condense the PMC passes.  usage: python profiles/install_r05.py [gpurun_out/r05]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
src = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "gpurun_out", "r05"))


def cp(a, b):
    shutil.copyfile(os.path.join(src, a), os.path.join(here, b))
    print(b)


cp("a/a_kernel_stats.csv", "r05_a_bench20_kernel_stats.csv")
cp("a_bench_line.json", "r05_a_bench20_line.json")
cp("b/b_kernel_stats.csv", "r05_b_isolated_kernel_stats.csv")
cp("b_isolated_line.json", "r05_b_isolated_line.json")
cp("c_bench_line.json", "r05_c_bench_line.json")
cp("e_cli_timing.log", "r05_e_cli_timing.log")
cp("e/e_kernel_stats.csv", "r05_e_cli_kernel_stats.csv")
cp("f_many_positions.txt", "r05_f_many_positions.txt")
for f in ("g_dist_emulated_line.json", "g_dist_worker_form_line.json", "g_dist_staged_form_line.json", "g_plain_4000_line.json"):
    cp(f, "r05_" + f)
cp("h/h_kernel_stats.csv", "r05_h_ingest_kernel_stats.csv")
cp("h_ingest.txt", "r05_h_ingest.txt")
subprocess.check_call([sys.executable, os.path.join(here, "make_pmc_traffic.py"), "r05_b", os.path.join(src, "pf"), os.path.join(src, "pw")])

# the ingest's kernels: one line per dispatch, then the averages
# (the first launch of cigar_runs and the first size of the planes kernel do the work on CCS reads; their second forms and the
# column-by-column kernel find nothing to do there and are listed beside them)
match = {"cigar_runs_kernel": "cigar_runs_kernel<64u", "ingest_planes_kernel": "4u, false>",
         "cigar_runs_kernel, second launch": "cigar_runs_kernel<512u", "ingest_planes_kernel, second size": "16u, true>"}
names = tuple(match)
avg = {n: {} for n in names}
for counter, d in (("FETCH_SIZE", "hf"), ("WRITE_SIZE", "hw")):
    rows = []
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for n in names:
                if match[n] in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), n, r["Grid_Size"], counter, float(r["Counter_Value"])))
    rows.sort()
    with open(os.path.join(here, f"r05_h_ingest_pmc_{counter}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Counter_Name", "Counter_Value_KB"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], f"{r[4]:.6f}"])
    for n in names:
        v = [r[4] for r in rows if r[1] == n]
        avg[n][counter] = sum(v) / max(1, len(v))
out = {"workload": "100000 reads x 3000 columns, 202.7 MB of records (150 MB packed bases, 51 MB cigar words, offsets), 112.9 MB of planes",
       "correction": "FETCH_SIZE x2 (gfx950 wide-stream correction: holds for 16-byte-per-lane streams; the cigar / run loads are narrower, "
                     "so their reads are an upper bound), WRITE_SIZE as read",
       "kernels": {n: {"FETCH_SIZE_KB_avg": avg[n]["FETCH_SIZE"], "WRITE_SIZE_KB_avg": avg[n]["WRITE_SIZE"],
                       "hbm_read_bytes(x2 corrected)": avg[n]["FETCH_SIZE"] * 2048.0, "hbm_write_bytes": avg[n]["WRITE_SIZE"] * 1024.0}
                   for n in names}}
json.dump(out, open(os.path.join(here, "r05_h_ingest_pmc_traffic.json"), "w"), indent=1)
print("r05_h_ingest_pmc_traffic.json")
for f, name in (("h_skip.txt", "r05_h_ingest_phases_off.txt"), ("h_sq.txt", "r05_h_ingest_sq_counters.txt"), ("h_stamps.txt", "r05_h_ingest_stamps.txt"), ("h_noisy.txt", "r05_h_ingest_indel_rich.txt")):
    if os.path.exists(os.path.join(src, f)):
        cp(f, name)
