# end-to-end `juliet` on the 100k x 3 kb synthetic BAM: stage timings and the kernels' durations (rocprofv3)
set -e
cd $GRAFT_REPO_ROOT
B=/tmp/e2e.bam; CFG=/tmp/e2e.json
[ -f $B ] || minorseq_amd/bin/juliet-synth --reads 100000 --cols 3000 --seed 2 -o $B --config-out $CFG
ls -la $B
for i in 1 2 3; do minorseq_amd/bin/juliet --timing -c $CFG --mode-phasing $B /tmp/e2e.out.json 2>&1 | grep timing; echo; done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cli_prof
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/cli_prof -o cli --output-format csv -- $GRAFT_REPO_ROOT/minorseq_amd/bin/juliet -c $CFG --mode-phasing $B /tmp/e2e.out2.json > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/cli_prof/cli_kernel_stats.csv | cut -c1-200
