# instructions and VALU-active cycles of ingest_planes_kernel with phases switched off (tuning build): which phase pays what per instruction
R=$GRAFT_REPO_ROOT
export JL_LIB=$R/tools_tuning/lib_exp/libjuliet_hip.so
cd /tmp && export TMPDIR=/tmp
for sk in ${SKIPS:-0 1 21 31 32 96}; do
  O=$R/gpurun_out/ingpmcskip/$sk; rm -rf $O; mkdir -p $O
  JL_ING_SKIP=$sk timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o p -- python3 $R/tools_tuning/ingest_time.py 100000 3000 4 > $O/out.txt 2> $O/err.txt || echo "skip=$sk failed"
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list); dur=[]
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ingest_planes" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ingest_planes" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
print("skip=$sk us(min under pmc)=%.0f " % (min(dur) if dur else -1) + " ".join(f"{c.replace('SQ_','')}={sum(v)/len(v)/1e6:.2f}M" for c, v in sorted(agg.items())), flush=True)
PY
done
