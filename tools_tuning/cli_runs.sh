# The juliet front end on a 100k-read rich-QV BAM, several runs per setting of the environment: when the decode ends, when the GPU
# context is ready, when the process is done (its own --timing laps).   usage: cli_runs.sh "<VAR=value ...>" ...   (on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
cd $R
B=/tmp/cli_qv.bam; CFG=/tmp/cli_qv.json
[ -f $B ] || minorseq_amd/bin/juliet-synth --reads 100000 --cols 3000 --seed 1000 --ref-seed 2 --rich-qv -o $B --config-out $CFG
minorseq_amd/bin/juliet -c $CFG --mode-phasing --min-qv 20 $B /tmp/cli.out.json > /dev/null 2>&1     # (the box's first process)
for v in "$@"; do
  for i in 1 2 3 4 5 6; do
    env $v minorseq_amd/bin/juliet --timing -c $CFG --mode-phasing --min-qv 20 $B /tmp/cli.out.json 2>&1 | python3 -c "
import sys, re
at = {}
for ln in sys.stdin:
    m = re.search(r'timing (\S.*?)\s+([0-9.]+) ms\s+\(at\s+([0-9.]+) ms\)', ln)
    if m: at[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)))
    m = re.search(r'gather ([0-9.]+) ms', ln)
    if m: g = float(m.group(1))
print('[$v] decode %.0f  context ready at %.0f  upload done at %.0f  (gather %.0f)  done at %.0f' % (at['bam decode'][1], at['context ready'][1], at['rest of the upload'][1], g, at['json / html'][1]))
"
  done
done
