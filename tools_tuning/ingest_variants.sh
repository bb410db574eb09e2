# the record ingest built with other tile / sweep shapes (tools_tuning/build_tuning_lib.sh "-DJL_INGEST_TILE=.. -DJL_INGEST_SWEEP=..u" lib_tT_sS.so):
# each checked against synth.rows on 20k reads, then timed on 100k x 3000 under rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export JL_LIB=$R/tools_tuning/lib_exp/$lib
  chk=$(python3 $R/tools_tuning/ingest_time.py 20000 3000 2 2>&1 | grep -c "matrix = synth.rows")
  O=$R/gpurun_out/ingvar/$lib; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools_tuning/ingest_time.py 100000 3000 6 > $O/out.txt 2> $O/err.txt
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "$lib verified=$chk $(grep ingest_planes $f | grep "4u, false" | awk -F, '{print "planes min", $(NF-2)}') $(grep "cigar_runs_kernel<64u" $f | awk -F, '{print "runs min", $(NF-2)}') $(grep 'builds:' $O/out.txt)"
done
