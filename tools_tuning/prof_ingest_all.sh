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
python3 $R/tools_tuning/ingest_summary.py $O > $O/summary.txt
cat $O/summary.txt
