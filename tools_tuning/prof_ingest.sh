# rocprofv3 kernel statistics of the record ingest alone (tools_tuning/ingest_time.py)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ing
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- python3 $R/tools_tuning/ingest_time.py "$@" > $O/out.txt 2> $O/err.txt
cat $O/out.txt
f=$(find $O/k -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-200
cp $f $O/kernel_stats.csv
