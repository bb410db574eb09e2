# cigar_runs_kernel alone (JL_ING_ONLY_RUNS: the launcher stops behind it), whole and with parts compiled out (-DJL_RUNS_PROBE=1: no
# descriptors, =2: no entries either): libs built by build_tuning_lib.sh into tools_tuning/lib_exp/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export JL_ING_ONLY_RUNS=1
for lib in libjuliet_hip.so lib_runs_probe1.so lib_runs_probe2.so; do
  export JL_LIB=$R/tools_tuning/lib_exp/$lib
  O=$R/gpurun_out/runsprobe/$lib; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools_tuning/ingest_time.py 100000 3000 6 > $O/out.txt 2> $O/err.txt
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "$lib $(grep 'cigar_runs_kernel<64u' $f | awk -F, '{print "runs avg", $(NF-4), "min", $(NF-2)}')"
done
