# The record ingest alone (tools_tuning/ingest_time.py) under rocprofv3: kernel statistics, then HBM traffic (FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md prescribes), then the SQ counters — everything under
# gpurun_out/<tag>/.   usage: prof_ingest_all.sh <tag> [sq] -- <ingest_time.py arguments>      (`sq`: the SQ passes too)
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SQ=0
if [ "$1" = "sq" ]; then SQ=1; shift; fi
[ "$1" = "--" ] && shift
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- python3 $R/tools_tuning/ingest_time.py "$@" > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
cat $O/out.txt
f=$(find $O/k -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
head -8 $O/kernel_stats.csv | cut -c1-220
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o p -- python3 $R/tools_tuning/ingest_time.py "$@" > /dev/null 2> $O/$c.err || { echo "$c pass failed"; tail -3 $O/$c.err; }
done
if [ $SQ = 1 ]; then
i=0
for set in "VALUBusy VALUUtilization MemUnitStalled OccupancyPercent" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq$i -o p -- python3 $R/tools_tuning/ingest_time.py "$@" > /dev/null 2> $O/sq$i.err || { echo "SQ set $i failed: $set"; tail -2 $O/sq$i.err; }
done
fi
python3 - <<PY > $O/summary.txt
import csv, glob, collections, json
O = "$O"
def short(n):
    for k in ("ingest_planes_kernel", "cigar_runs_kernel", "cigar_walk_kernel"):
        if k in n:
            return k + n[n.index("<"):n.index(">") + 1] if "<" in n else k
    return None
print("kernel statistics (rocprofv3 --kernel-trace --stats):")
for r in csv.DictReader(open(O + "/kernel_stats.csv")):
    if short(r["Name"]):
        print(f"   {short(r['Name']):50s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}")
traffic = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(O + f"/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        traffic.setdefault(k, {})[c + "_KB_avg"] = sum(v) / len(v)
        traffic[k]["dispatches"] = len(v)
print("HBM traffic per dispatch (KB as the counters give them; FETCH_SIZE counts 64-byte requests as 32 on gfx950 wide streams: see MI355X_MICROARCH.md):")
print(json.dumps(traffic, indent=1))
json.dump(traffic, open(O + "/pmc_traffic_raw.json", "w"), indent=1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {sum(v)/len(v):18.1f}  ({len(v)} dispatches)")
PY
cat $O/summary.txt
