# A/B of two builds of the front end on one box, alternating: minorseq_amd/bin/juliet_old (built from another revision of
# juliet_main.cpp with the flags of minorseq_amd/host/Makefile; the A/B is skipped without it) against minorseq_amd/bin/juliet
cd $GRAFT_REPO_ROOT
B=/tmp/e2e.bam; CFG=/tmp/e2e.json
[ -f $B ] || minorseq_amd/bin/juliet-synth --reads 100000 --cols 3000 --seed 2 -o $B --config-out $CFG
show() { grep -E "timing (bam|context|rest|  uploader|device|plan|json)"; }
for i in 1 2 3 4 5; do
  if [ -x minorseq_amd/bin/juliet_old ]; then echo "== juliet_old"; minorseq_amd/bin/juliet_old --timing -c $CFG --mode-phasing $B /tmp/e2e.out.json 2>&1 | show; fi
  echo "== juliet"; minorseq_amd/bin/juliet --timing -c $CFG --mode-phasing $B /tmp/e2e.out.json 2>&1 | show
done
