# bench.py's once-through legs and resident replay under libraries that differ in -D flags (JL_LIB), alternating, two rounds.
# usage: bench_ab.sh "<flags A>" "<flags B>" ...   (on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
cd $R
i=0
for f in "$@"; do
  i=$((i+1))
  bash tools_tuning/build_tuning_lib.sh "$f" libjuliet_v$i.so > /dev/null 2>&1 || { echo "build $i failed: $f"; exit 1; }
done
for round in $(seq 1 ${ROUNDS:-2}); do
  i=0
  for f in "$@"; do
    i=$((i+1))
    JL_LIB=$R/tools_tuning/lib_exp/libjuliet_v$i.so python3 bench.py --steps ${STEPS:-16000} --no-config3 --no-cpu-baseline --no-end-to-end ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
q,o=d['once_through_qv'],d['once_through']
print('[$f] replay %.2f us | qv step %.1f ingest %.1f | plain step %.1f ingest %.1f' % (1e3*d['ms_per_step'], 1e3*q['ms_per_step'], 1e3*q['ingest_ms'], 1e3*o['ms_per_step'], 1e3*o['ingest_ms']))
"
  done
done
