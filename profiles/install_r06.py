"""Copy what profiles/collect_r06.sh left under gpurun_out/r06 into profiles/r06_* (the files the README table names) and condense
the PMC passes.  usage: python profiles/install_r06.py [gpurun_out/r06]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
src = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "gpurun_out", "r06"))


def cp(a, b):
    if not os.path.exists(os.path.join(src, a)):
        print("missing:", a)
        return
    shutil.copyfile(os.path.join(src, a), os.path.join(here, b))
    print(b)


cp("a/a_kernel_stats.csv", "r06_a_bench20_kernel_stats.csv")
cp("a_bench_line.json", "r06_a_bench20_line.json")
cp("b/b_kernel_stats.csv", "r06_b_isolated_kernel_stats.csv")
cp("b_isolated_line.json", "r06_b_isolated_line.json")
cp("c_bench_line.json", "r06_c_bench_line.json")
cp("e_cli_timing.log", "r06_e_cli_timing.log")
cp("e/e_kernel_stats.csv", "r06_e_cli_kernel_stats.csv")
cp("f_many_positions.txt", "r06_f_many_positions.txt")
for f in ("g_dist_emulated_line.json", "g_dist_staged_form_line.json", "g_plain_4000_line.json"):
    cp(f, "r06_" + f)
cp("i_fold_ab.txt", "r06_i_fold_ab.txt")
cp("j_phases_off.txt", "r06_j_ingest_phases_off.txt")
cp("j_workgroups_per_cu.txt", "r06_j_ingest_workgroups_per_cu.txt")
if os.path.isdir(os.path.join(src, "pf")) and os.path.isdir(os.path.join(src, "pw")):
    subprocess.check_call([sys.executable, os.path.join(here, "make_pmc_traffic.py"), "r06_b", os.path.join(src, "pf"), os.path.join(src, "pw")])

# the ingest, both input shapes: the kernel statistics, the summary (statistics, raw PMC averages, SQ counters) and the HBM traffic
shapes = {"h": ("r06_h_ingest", "100000 reads x 3000 columns, filtered bases as N letters (127 cigar ops a read): 202.7 MB of records (150 MB packed bases, "
                                "51 MB cigar words, offsets), 112.9 MB of planes"),
          "h_qv": ("r06_h_qv", "100000 reads x 3000 columns as `ccs --richQVs` leaves them (ten cigar ops a read, one quality byte per base, min_qv 20): "
                               "456.2 MB of records (150 MB packed bases, 300 MB qualities, 4 MB cigar words, offsets), 112.9 MB of planes")}
for d, (tag, workload) in shapes.items():
    cp(f"{d}/kernel_stats.csv", f"{tag}_kernel_stats.csv")
    cp(f"{d}/summary.txt", f"{tag}_summary.txt")
    cp(f"{d}/out.txt", f"{tag}.txt")
    cp(f"{d}_one/kernel_stats.csv", f"{tag}_one_stream_kernel_stats.csv")
    cp(f"{d}_one/out.txt", f"{tag}_one_stream.txt")
    raw = os.path.join(src, d, "pmc_traffic_raw.json")
    if os.path.exists(raw):
        t = json.load(open(raw))
        out = {"workload": workload,
               "correction": "FETCH_SIZE x2 (gfx950 wide-stream correction: holds for 16-byte-per-lane streams; narrower loads — cigar words, "
                             "entries — make it an upper bound), WRITE_SIZE as read; KB as the counters give them",
               "kernels": {k: dict(v, **{"hbm_read_bytes(x2 corrected)": v.get("FETCH_SIZE_KB_avg", 0.0) * 2048.0,
                                         "hbm_write_bytes": v.get("WRITE_SIZE_KB_avg", 0.0) * 1024.0}) for k, v in t.items()}}
        json.dump(out, open(os.path.join(here, f"{tag}_pmc_traffic.json"), "w"), indent=1)
        print(f"{tag}_pmc_traffic.json")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in glob.glob(os.path.join(src, d, c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if ("ingest_" in r["Kernel_Name"] or "cigar_" in r["Kernel_Name"]) and r["Counter_Name"] == c:
                    n = r["Kernel_Name"]
                    rows.append((int(r["Dispatch_Id"]), n[n.index("::") + 2:n.index("(")] if "::" in n and "(" in n else n[:60], r["Grid_Size"], c, float(r["Counter_Value"])))
        if rows:
            rows.sort()
            with open(os.path.join(here, f"{tag}_pmc_{c}.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Counter_Name", "Counter_Value_KB"])
                for r in rows:
                    w.writerow([r[0], r[1], r[2], r[3], f"{r[4]:.6f}"])
            print(f"{tag}_pmc_{c}.csv")
