set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ingpmc2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/hf -o hf -- python3 $R/tools_tuning/ingest_time.py 100000 3000 8 > /dev/null 2> $O/hf.err
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list); dur=collections.defaultdict(list)
for f in glob.glob("$O/hf/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        for name in ("ingest_planes_kernel","cigar_runs_kernel"):
            if name in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": agg[name].append(float(r["Counter_Value"]))
for f in glob.glob("$O/hf/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        for name in ("ingest_planes_kernel","cigar_runs_kernel"):
            if name in r["Kernel_Name"]: dur[name].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))//1000)
for k,v in agg.items(): print(k, "FETCH_SIZE KB avg", sum(v)/len(v), "durations us", dur[k])
PY
