# Round-2 profile collection on the MI355X box (run through gpurun from the repo root):
#   1. the driver's command under rocprofv3 --kernel-trace --stats            -> r02_a_bench20_*
#   2. the dominant kernel alone (one population)                              -> r02_b_isolated_*
#   3. PMC passes FETCH_SIZE / WRITE_SIZE over the isolated loop (separate runs, --kernel-trace only)  -> r02_b_pmc_*
#   4. bench.py defaults, unprofiled                                           -> r02_c_bench_line.json
#   5. the juliet front end on a 100k-read BAM                                 -> r02_cli_*
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o a -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/a_bench_line.json 2> $O/a.err
echo "1 done"; 
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o b -- python3 $R/profiles/isolated_pileup.py 2000 > $O/b_isolated_line.json 2> $O/b.err
echo "2 done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o pf -- python3 $R/profiles/isolated_pileup.py 25 > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o pw -- python3 $R/profiles/isolated_pileup.py 25 > /dev/null 2> $O/pw.err
echo "3 done"
cd $R
python3 bench.py > $O/c_bench_line.json 2> $O/c.err
python3 profiles/isolated_pileup.py 2000 > $O/c_isolated_unprofiled.json 2>> $O/c.err
echo "4 done"
bash tools_tuning/cli_profile.sh > $O/cli_timing.log 2>&1 || true
cp -r gpurun_out/cli_prof $O/cli_prof 2>/dev/null || true
echo "5 done"
ls $O
