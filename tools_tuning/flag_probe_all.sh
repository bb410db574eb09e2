# Every ingest kernel's time (alone on the device: one stream) under builds of the library that differ in -D flags, with qualities.
# usage: flag_probe_all.sh "<flags>" ...   (on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
cd $R
export JL_ING_ONE_STREAM=1
for v in "$@"; do
  bash tools_tuning/build_tuning_lib.sh "$v" libjuliet_fl.so > /dev/null 2>&1 || { echo "build '$v' failed"; continue; }
  ok=$(JL_LIB=$R/tools_tuning/lib_exp/libjuliet_fl.so python3 tools_tuning/ingest_time.py 20000 3000 2 ${Q:-20} 2>&1 | grep -c "matrix = synth.rows")
  JL_LIB=$R/tools_tuning/lib_exp/libjuliet_fl.so bash tools_tuning/prof_ingest.sh 100000 3000 12 ${Q:-20} > /dev/null 2>&1
  python3 - <<PY
import csv
t = {}
for r in csv.DictReader(open("gpurun_out/ing/kernel_stats.csv")):
    n = r["Name"]
    for k in ("ingest_planes_kernel<true, 4u", "ingest_planes_kernel<false, 4u", "qual_mask_kernel", "cigar_walk_kernel", "cigar_runs_kernel", "16u, true"):
        if k in n: t[k] = float(r["AverageNs"]) / 1e3
print("[$v] cells ok: $ok |", " | ".join("%s %.1f" % (k.replace("ingest_planes_kernel", "planes"), v) for k, v in t.items()), "| sum %.1f" % sum(t.values()))
PY
done
