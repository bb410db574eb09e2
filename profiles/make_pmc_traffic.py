"""Condense the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<tag>_pmc_*.csv and pmc_traffic.json.

usage: python profiles/make_pmc_traffic.py <tag> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>
The passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 profiles/isolated_pileup.py 25
            (and the same with --pmc WRITE_SIZE): the dominant kernel alone, every dispatch attributed
Correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64-byte units of the wide stream twice too low on
gfx950 -> x2; WRITE_SIZE as read; both in KB."""
import csv
import glob
import json
import os
import sys

tag, d_fetch, d_write = sys.argv[1:4]
here = os.path.dirname(os.path.abspath(__file__))


def rows(d, counter):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in ("pileup_group_kernel", "pileup_planes_group_kernel", "pileup_fold_group_kernel")) and r["Counter_Name"] == counter:
                out.append(r)
    return out


def condense(rs, counter):
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    path = os.path.join(here, f"{tag}_pmc_{counter}.csv")
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(keep)
        for r in rs:
            name = ("pileup_planes_group_kernel<3,4>" if "planes_group_kernel<3" in r["Kernel_Name"] else
                    "pileup_fold_group_kernel<3,4>" if "fold_group_kernel<3" in r["Kernel_Name"] else r["Kernel_Name"][:60])
            w.writerow([r["Dispatch_Id"], name] + [r[k] for k in keep[2:]])
    # full launches only (the set-up pass also launches partial groups)
    full = max(int(r["Grid_Size"]) for r in rs)
    vals = [float(r["Counter_Value"]) for r in rs if int(r["Grid_Size"]) == full]
    return sum(vals) / len(vals), len(vals), full


fa, nf, grid = condense(rows(d_fetch, "FETCH_SIZE"), "FETCH_SIZE")
wa, nw, _ = condense(rows(d_write, "WRITE_SIZE"), "WRITE_SIZE")
windows = grid // 256 // 1000          # 1000 chunks (workgroups) per 3000-column window
alg = windows * 100_000 * 3000 * 3 // 8   # 3 bits per cell: every cell of the resident planes once
hbm = int(round((2.0 * fa + wa) * 1024))
old = json.load(open(os.path.join(here, "pmc_traffic.json")))
new = {
    "kernel": "pileup_fold_group_kernel<3,4> (round 6: the pileup with the Fisher stage of its codons in the epilogue; rounds 4-5: pileup_planes_group_kernel<3,4>)",
    "layout": "bit planes: 3 bits per cell, the one resident format (112.5 MB of cells per 100k x 3000 window; the library pads a plane to whole 128-byte lines: 112.9 MB allocated)",
    "windows_per_launch": windows,
    "workload": f"{windows} windows of 100000 reads x 3000 columns per launch (bench default: --group {windows})",
    "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa, "launches_averaged": [nf, nw],
    "correction": "FETCH_SIZE x2 (gfx950 wide-stream correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; unit KB",
    "pileup_kernel_hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": alg,
    "ratio": hbm / alg,
    "source": f"profiles/{tag}_pmc_FETCH_SIZE.csv, profiles/{tag}_pmc_WRITE_SIZE.csv (separate --pmc passes with --kernel-trace only over "
              "profiles/isolated_pileup.py; profiles/make_pmc_traffic.py)",
}
for k in ("four_window_launch", "single_window_kernel"):
    if k in old:
        new[k + "_rounds_1_2_nibble_kernel"] = old[k]
json.dump(new, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(new, indent=1))
