# Sibling tiles a workgroup of the planes kernel (JL_INGEST_NT / _NT_QV) and the qualities asked for ahead (JL_INGEST_QUAL_AHEAD):
# builds of the library, the planes kernel timed alone on one stream, 20 000 reads checked cell by cell first.   (run on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
cd $R
export JL_ING_ONE_STREAM=1
for v in "${@:-1 1 7}"; do
  set -- $v
  bash tools_tuning/build_tuning_lib.sh "-DJL_INGEST_NT=$1 -DJL_INGEST_NT_QV=$2 -DJL_INGEST_QUAL_AHEAD=$3" libjuliet_nt.so > /dev/null 2>&1 || { echo "build $v failed"; continue; }
  for q in 0 20; do
    JL_LIB=$R/tools_tuning/lib_exp/libjuliet_nt.so python3 tools_tuning/ingest_time.py 20000 3000 2 $q 2>&1 | grep -c "matrix = synth.rows"
    JL_LIB=$R/tools_tuning/lib_exp/libjuliet_nt.so bash tools_tuning/prof_ingest.sh 100000 3000 12 $q > /dev/null 2>&1
    python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ing/kernel_stats.csv")):
    n = r["Name"]
    if "ingest_planes_kernel" in n and "4u, false" in n:
        print("NT $1 NT_QV $2 ahead $3, min_qv $q:", n[n.index("<"):n.index(">") + 1], "avg %.1f min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  done
done
