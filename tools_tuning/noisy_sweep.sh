# the record ingest on reads with more and more indels (ingest_noisy.py), the kernels' times from rocprofv3; then the clean case
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in "0 0" "2500 0.0025" "5000 0.005" "10000 0.01" "20000 0.02"; do
  set -- $c
  O=$R/gpurun_out/noisy/i$1; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools_tuning/ingest_noisy.py 100000 3000 $1 $2 > $O/out.txt 2> $O/err.txt
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "ins_ppm $1 del $2: $(grep 'ops per read\|equal' $O/out.txt | tr '\n' ' ')"
  grep "ingest_\|cigar_runs" $f | awk -F, '{print "   ", $1, "calls", $2, "avg ns", $4, "min", $(NF-2)}' | cut -c1-200
done
O=$R/gpurun_out/noisy/clean; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools_tuning/ingest_time.py 100000 3000 6 > $O/out.txt 2> $O/err.txt
f=$(find $O -name "*kernel_stats.csv" | head -1)
echo "clean (ingest_time.py 100000 3000): $(grep 'builds:' $O/out.txt)"
grep "ingest_\|cigar_runs" $f | awk -F, '{print "   ", $1, "calls", $2, "avg ns", $4, "min", $(NF-2)}' | cut -c1-200
