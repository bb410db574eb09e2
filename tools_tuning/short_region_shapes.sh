# the driver's command (--steps 20 --warmup 5) with other launch shapes: windows per launch x launches in flight
# usage: short_region_shapes.sh "G N" "G N" ...   (JL_LIB: another build of the library)
F="--steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-config3 --no-config4 --no-once-through"
for shape in "$@"; do
  set -- $shape
  for rep in 1 2 3; do
    python3 bench.py $F --group $1 --inflight $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $1 inflight $2:', round(1000*d['ms_per_step'],2), 'us/step')"
  done
done
