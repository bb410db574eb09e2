# PMC counters of the ingest kernels (one pass, --kernel-trace only besides --pmc)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ingpmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $R/tools_tuning/ingest_time.py 100000 3000 6 > $O/out1.txt 2> $O/err1.txt || { echo "failed"; tail -3 $O/err1.txt; }
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "ingest_planes" if "ingest_planes" in r["Kernel_Name"] else "cigar_runs" if "cigar_runs" in r["Kernel_Name"] else None
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "ingest_planes" if "ingest_planes" in r["Kernel_Name"] else "cigar_runs" if "cigar_runs" in r["Kernel_Name"] else None
        if k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    print(k, "duration us (under pmc):", [x // 1000 for x in dur[k]])
    for c, v in sorted(d.items()):
        print(f"   {c:26s} {sum(v)/len(v):16.0f}  ({len(v)} dispatches)")
PY
