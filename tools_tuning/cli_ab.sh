# A/B of two builds of the front end on one box: minorseq_amd/bin/juliet_old against minorseq_amd/bin/juliet, alternating
cd $GRAFT_REPO_ROOT
B=/tmp/e2e.bam; CFG=/tmp/e2e.json
[ -f $B ] || minorseq_amd/bin/juliet-synth --reads 100000 --cols 3000 --seed 2 -o $B --config-out $CFG
for i in 1 2 3 4; do
  for w in juliet_old juliet; do
    echo "== $w"; minorseq_amd/bin/$w --timing -c $CFG --mode-phasing $B /tmp/e2e.out.json 2>&1 | grep -E "timing (bam|context|rest|  uploader|device|plan|json)"
  done
done
