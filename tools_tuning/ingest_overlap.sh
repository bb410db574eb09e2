# does a planes kernel that leaves LDS for a cigar_runs workgroup let the next build's cigar_runs run beside it?  (two streams)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export JL_LIB=$R/tools_tuning/lib_exp/$lib
  O=$R/gpurun_out/ingoverlap/$lib; rm -rf $O; mkdir -p $O
  python3 $R/tools_tuning/ingest_time.py 100000 3000 40 > $O/plain.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools_tuning/ingest_time.py 100000 3000 40 > $O/out.txt 2> $O/err.txt
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "$lib: $(grep builds: $O/plain.txt) | $(grep "4u, false\|3u, false" $f | awk -F, '{print "planes avg", $(NF-4), "min", $(NF-2)}') $(grep 'cigar_runs_kernel<64u\|cigar_runs_kernel<32u' $f | awk -F, '{print "runs avg", $(NF-4), "min", $(NF-2)}')"
done
