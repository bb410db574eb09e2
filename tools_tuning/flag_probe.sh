# The planes kernel under builds of the library that differ in -D flags: timed alone on one stream under rocprofv3, without and with
# qualities, 20 000 reads checked cell by cell first.   usage: flag_probe.sh "<flags>" "<flags>" ...   ("" = the defaults; on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
cd $R
export JL_ING_ONE_STREAM=1
for v in "$@"; do
  bash tools_tuning/build_tuning_lib.sh "$v" libjuliet_fl.so > /dev/null 2>&1 || { echo "build '$v' failed"; continue; }
  for q in ${QS:-0 20}; do
    JL_LIB=$R/tools_tuning/lib_exp/libjuliet_fl.so python3 tools_tuning/ingest_time.py 20000 3000 2 $q 2>&1 | grep -c "matrix = synth.rows"
    for rep in 1 2; do
      JL_LIB=$R/tools_tuning/lib_exp/libjuliet_fl.so bash tools_tuning/prof_ingest.sh 100000 3000 12 $q > /dev/null 2>&1
      python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ing/kernel_stats.csv")):
    n = r["Name"]
    if "ingest_planes_kernel" in n and "4u, false" in n:
        print("[$v] min_qv $q:", n[n.index("<"):n.index(">") + 1], "avg %.1f min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
    done
  done
done
