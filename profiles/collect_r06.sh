# Round-6 profile collection on the MI355X box (through gpurun from the repo root: bash profiles/collect_r06.sh <part> ...; a call
# takes the parts it is given, in order — the whole collection does not fit one gpurun call):
#   a  the driver's command under rocprofv3 --kernel-trace --stats                                        -> r06_a_bench20_*
#   b  the dominant kernel alone (one population) + PMC passes FETCH_SIZE / WRITE_SIZE (separate runs)    -> r06_b_*
#   c  bench.py defaults, unprofiled (end_to_end, once_through, once_through_qv, configs[3]/[4], CPU)     -> r06_c_bench_line.json
#   e  the juliet front end: rich-QV BAM with --min-qv 20 (what bench.py's end_to_end runs) and the plain BAM of round 5: --timing x3,
#      one run under rocprofv3                                                                            -> r06_e_cli_*
#   f  a window with sixteen variant positions + one window alone                                        -> r06_f_many_positions.txt
#   g  one-rank emulation of the N > 1 step loop (bound / staged / plain)                                 -> r06_g_*
#   h  the record ingest alone, both input shapes: kernel stats, PMC FETCH_SIZE / WRITE_SIZE, SQ counters  -> r06_h_ingest_*, r06_h_qv_*
#   i  the Fisher stage folded into the pileup launch against the separate call launch, alternating      -> r06_i_fold_ab.txt
#   j  where the ingest's time goes: phases switched off, workgroups a CU (tuning builds)                 -> r06_j_*
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
for part in "$@"; do
case $part in
a)
  cd /tmp; rm -rf $O/a
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o a -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/a_bench_line.json 2> $O/a.err
  ;;
b)
  cd /tmp; rm -rf $O/b $O/pf $O/pw
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o b -- python3 $R/profiles/isolated_pileup.py 2000 > $O/b_isolated_line.json 2> $O/b.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o pf -- python3 $R/profiles/isolated_pileup.py 25 > /dev/null 2> $O/pf.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o pw -- python3 $R/profiles/isolated_pileup.py 25 > /dev/null 2> $O/pw.err
  ;;
c)
  cd $R; python3 bench.py > $O/c_bench_line.json 2> $O/c.err
  ;;
e)
  cd $R
  B=/tmp/e2e_qv.bam; CFG=/tmp/e2e_qv.json; P=/tmp/e2e.bam; PCFG=/tmp/e2e.json
  minorseq_amd/bin/juliet-synth --reads 100000 --cols 3000 --seed 1000 --ref-seed 2 --rich-qv -o $B --config-out $CFG
  minorseq_amd/bin/juliet-synth --reads 100000 --cols 3000 --seed 2 -o $P --config-out $PCFG
  { echo "rich-QV BAM ($(stat -c %s $B) bytes), --min-qv 20:"
    for i in 1 2 3; do minorseq_amd/bin/juliet --timing -c $CFG --mode-phasing --min-qv 20 $B /tmp/e2e.out.json 2>&1 | grep timing; echo; done
    echo "plain BAM ($(stat -c %s $P) bytes; filtered bases as N letters), as in round 5:"
    for i in 1 2 3; do minorseq_amd/bin/juliet --timing -c $PCFG --mode-phasing $P /tmp/e2e.out.json 2>&1 | grep timing; echo; done; } > $O/e_cli_timing.log
  cd /tmp; rm -rf $O/e
  JL_SLOW_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e -o e -- $R/minorseq_amd/bin/juliet -c $CFG --mode-phasing --min-qv 20 $B /tmp/e2e.out2.json > /dev/null 2> $O/e.err
  ;;
f)
  cd $R
  python3 tools_tuning/generic_phase_cost.py --check > $O/f_many_positions.txt 2> $O/f.err
  python3 tools_tuning/one_window_latency.py >> $O/f_many_positions.txt 2>> $O/f.err
  ;;
g)
  cd $R
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 JL_BENCH_FORCE_DIST=1 python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-once-through > $O/g_dist_emulated_line.json 2> $O/g.err
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 JL_BENCH_FORCE_DIST=1 JL_EXCHANGE_STAGED=1 python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-once-through > $O/g_dist_staged_form_line.json 2>> $O/g.err
  python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-once-through --no-end-to-end > $O/g_plain_4000_line.json 2>> $O/g.err
  ;;
h)
  cd $R
  bash tools_tuning/prof_ingest_all.sh r06/h sq -- 100000 3000 16 0 > $O/h.log 2>&1 || true
  bash tools_tuning/prof_ingest_all.sh r06/h_qv sq -- 100000 3000 16 20 > $O/h_qv.log 2>&1 || true
  JL_ING_ONE_STREAM=1 bash tools_tuning/prof_ingest_all.sh r06/h_one -- 100000 3000 16 0 > $O/h_one.log 2>&1 || true
  JL_ING_ONE_STREAM=1 bash tools_tuning/prof_ingest_all.sh r06/h_qv_one -- 100000 3000 16 20 > $O/h_qv_one.log 2>&1 || true
  ;;
i)
  cd $R; bash tools_tuning/fold_ab.sh 3 > $O/i_fold_ab.txt 2>&1
  ;;
j)
  cd $R
  bash tools_tuning/build_tuning_lib.sh > /dev/null 2>&1
  { export JL_ING_ONE_STREAM=1
    echo "planes kernel alone on one stream, us (JL_ING_SKIP bits: 1 no bases staged (no quality loads), 2 no table, 4 no plane stores, 16 no transposing, 64 the"
    echo "bases taken as they are (no conversion, no quality loads), 128 16-byte stores by one lane of four (the same requests), 256 non-temporal stores,"
    echo "2048 the same bytes as 64-byte requests (a quarter of the requests; wrong data by design))"
    echo "with qualities (min_qv 20):"; MIN_QV=20 SKIPS="0 1 2 4 16 64 128 256 2048" bash tools_tuning/skip_ingest.sh
    echo "without:"; MIN_QV=0 SKIPS="0 1 2 4 16 64 128 256 2048" bash tools_tuning/skip_ingest.sh; } > $O/j_phases_off.txt 2>&1 || true
  bash tools_tuning/lds_sensitivity.sh > $O/j_workgroups_per_cu.txt 2>&1 || true
  ;;
esac
echo "$part done"
done
