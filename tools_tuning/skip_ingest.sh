# time of the ingest kernels with phases switched off (tuning build; results are wrong by design)
# JL_ING_SKIP bits: 0 no bases, 1 no deletions, 2 no plane stores, 3 no second pass, 4 no transposing, 5 codes made but not placed,
# 6 no conversion (the packed bases taken as codes)
R=$GRAFT_REPO_ROOT
export JL_LIB=$R/tools_tuning/lib_exp/libjuliet_hip.so
cd /tmp && export TMPDIR=/tmp
for sk in ${SKIPS:-0 1 2 3 4 8 16 20 21 23 31}; do
  O=$R/gpurun_out/ingskip/$sk; rm -rf $O; mkdir -p $O
  JL_ING_SKIP=$sk rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools_tuning/ingest_time.py 100000 3000 6 ${MIN_QV:-0} > $O/out.txt 2> $O/err.txt
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "skip=$sk $(grep ingest_planes $f | grep "4u, false" | awk -F, '{print "planes avg", $(NF-4), "min", $(NF-2)}') $(grep "cigar_runs_kernel<64u" $f | awk -F, '{print "runs min", $(NF-2)}')"
done
