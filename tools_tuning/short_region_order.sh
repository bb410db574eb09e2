# the 20-step region with the short launch first / last and with launches of four windows (round 5; the switch JL_BENCH_REM_FIRST
# lived in bench.py for this measurement only: first 36.4-37.1, last 25.0-25.6, G = 4: 25.5-26.6 / 38.8-39.3 us per step)
F="--steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-config4 --no-once-through"
for rep in 1 2 3; do
for cfg in "1 8 4" "0 8 4" "1 4 4" "1 4 6"; do set -- $cfg
  JL_BENCH_REM_FIRST=$1 python3 bench.py $F --group $2 --inflight $3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('rem_first=$1 G=$2 inflight=$3:',round(1000*d['ms_per_step'],2),'us/step; frac',round(d['roofline']['frac'],3))"
done; done
