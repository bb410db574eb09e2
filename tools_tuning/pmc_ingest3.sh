# what the record ingest's kernels keep busy: derived metrics + raw unit counters, one rocprofv3 --pmc pass per group
# (SQ counters and derived metrics only: passes with GRBM_*, TA_* / TD_*, TCP_* or TCC_* counters never came back on this pool;
# FETCH_SIZE / WRITE_SIZE alone do: pmc_ingest2.sh)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ingpmc3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "VALUBusy VALUUtilization MemUnitStalled OccupancyPercent" "MeanOccupancyPerActiveCU LDSBankConflict" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools_tuning/ingest_time.py 100000 3000 4 > $O/out$i.txt 2> $O/err$i.txt || { echo "set $i failed: $set"; tail -2 $O/err$i.txt; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]   # (the first size / the first launch: the second forms find nothing to do on these reads)
        k = "ingest_planes" if ("ingest_planes" in n and "4u, false>" in n) else "cigar_runs" if "cigar_runs_kernel<64u" in n else None
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {sum(v)/len(v):18.1f}  ({len(v)} dispatches)")
PY
