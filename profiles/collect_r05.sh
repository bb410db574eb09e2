# Round-5 profile collection on the MI355X box (run through gpurun from the repo root: bash profiles/collect_r05.sh):
#   a  the driver's command under rocprofv3 --kernel-trace --stats                                   -> r05_a_bench20_*
#   b  the dominant kernel alone (one population) + PMC passes FETCH_SIZE / WRITE_SIZE (separate runs) -> r05_b_*
#   c  bench.py defaults, unprofiled                                                                  -> r05_c_bench_line.json
#   e  the juliet front end on a 100k-read BAM: --timing x3, --windows 8, kernel trace (no planes_kernel) -> r05_e_cli_*
#   f  a window with sixteen variant positions (two-word fused launch) + one window alone             -> r05_f_many_positions.txt
#   g  one-rank emulation of the N > 1 step loop: exchange carried by the launch (default), worker-thread form, staged form, plain -> r05_g_*
#   h  the record ingest alone: kernel stats + PMC FETCH_SIZE / WRITE_SIZE of its kernels            -> r05_h_ingest_*
#      + the tuning build: phases switched off (times), SQ instruction counters, wall-clock stamps of the phases of sampled workgroups
#      + reads with 0 .. 2 % of indels (the planes kernel's second size, cigar_runs' second launch, the column-by-column kernel)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o a -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/a_bench_line.json 2> $O/a.err
echo "a done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o b -- python3 $R/profiles/isolated_pileup.py 2000 > $O/b_isolated_line.json 2> $O/b.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o pf -- python3 $R/profiles/isolated_pileup.py 25 > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o pw -- python3 $R/profiles/isolated_pileup.py 25 > /dev/null 2> $O/pw.err
echo "b done"
cd $R
python3 bench.py > $O/c_bench_line.json 2> $O/c.err
echo "c done"
B=/tmp/e2e.bam; CFG=/tmp/e2e.json
minorseq_amd/bin/juliet-synth --reads 100000 --cols 3000 --seed 2 -o $B --config-out $CFG
{ for i in 1 2 3; do minorseq_amd/bin/juliet --timing -c $CFG --mode-phasing $B /tmp/e2e.out.json 2>&1 | grep timing; echo; done
  echo "--windows 8:"; minorseq_amd/bin/juliet --timing -c $CFG --mode-phasing --windows 8 $B /tmp/e2e.w8.json 2>&1 | grep timing; } > $O/e_cli_timing.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/e -o e -- $R/minorseq_amd/bin/juliet -c $CFG --mode-phasing $B /tmp/e2e.out2.json > /dev/null 2> $O/e.err
cd $R
echo "e done"
python3 tools_tuning/generic_phase_cost.py --check > $O/f_many_positions.txt 2> $O/f.err
python3 tools_tuning/one_window_latency.py >> $O/f_many_positions.txt 2>> $O/f.err
echo "f done"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 JL_BENCH_FORCE_DIST=1 python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config4 --no-once-through > $O/g_dist_emulated_line.json 2> $O/g.err
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 JL_BENCH_FORCE_DIST=1 JL_BENCH_EXCHANGE=worker python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-config4 --no-once-through > $O/g_dist_worker_form_line.json 2>> $O/g.err
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 JL_BENCH_FORCE_DIST=1 JL_EXCHANGE_STAGED=1 python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-config4 --no-once-through > $O/g_dist_staged_form_line.json 2>> $O/g.err
python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-once-through > $O/g_plain_4000_line.json 2>> $O/g.err
echo "g done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/h -o h -- python3 $R/tools_tuning/ingest_time.py 100000 3000 40 > $O/h_ingest.txt 2> $O/h.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/hf -o hf -- python3 $R/tools_tuning/ingest_time.py 100000 3000 8 > /dev/null 2> $O/hf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/hw -o hw -- python3 $R/tools_tuning/ingest_time.py 100000 3000 8 > /dev/null 2> $O/hw.err
SKIPS="0 1 2 8 16 4 3 19" bash $R/tools_tuning/skip_ingest.sh > $O/h_skip.txt 2>&1 || true
bash $R/tools_tuning/pmc_ingest3.sh > $O/h_sq.txt 2>&1 || true
JL_ING_STAMPS=1 JL_LIB=$R/tools_tuning/lib_exp/libjuliet_hip.so python3 $R/tools_tuning/ingest_time.py 100000 3000 2 > $O/h_stamps.txt 2>&1 || true
bash $R/tools_tuning/noisy_sweep.sh > $O/h_noisy.txt 2>&1 || true
echo "h done"
find $O -name "*.csv" | head -40
