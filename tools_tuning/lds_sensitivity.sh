# How does the planes kernel's time depend on the workgroups a CU holds?  Builds of the library whose planes workgroups hold extra LDS
# (-DJL_INGEST_LDS_PAD: 5 -> 4 -> 3 -> 2 workgroups a CU), timed alone on one stream.   (run on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
cd $R
export JL_ING_ONE_STREAM=1
for pad in 0 8192 22000 48000; do
  bash tools_tuning/build_tuning_lib.sh "-DJL_INGEST_LDS_PAD=$pad" libjuliet_pad$pad.so > /dev/null 2>&1 || { echo "build $pad failed"; continue; }
  for q in 0 20; do
    JL_LIB=$R/tools_tuning/lib_exp/libjuliet_pad$pad.so bash tools_tuning/prof_ingest.sh 100000 3000 12 $q > /dev/null 2>&1
    python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ing/kernel_stats.csv")):
    n = r["Name"]
    if "ingest_planes_kernel" in n and "4u, false" in n:
        print("pad $pad qv $q:", n[n.index("ingest_planes"):n.index(">") + 1], "avg %.1f min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  done
done
