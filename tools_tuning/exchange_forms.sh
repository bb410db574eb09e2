# the one-rank emulation of the weak-scaling step with the exchange carried by the launch, with the worker-thread form, and
# the plain step, 4000 steps each
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
O=gpurun_out/xforms; mkdir -p $O
F="--steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-config4 --no-once-through"
MASTER_PORT=29517 JL_BENCH_FORCE_DIST=1 python3 bench.py $F > $O/bound.json 2> $O/bound.err || exit 1
MASTER_PORT=29518 JL_BENCH_FORCE_DIST=1 JL_BENCH_EXCHANGE=worker python3 bench.py $F > $O/worker.json 2> $O/worker.err || exit 1
MASTER_PORT=29519 JL_BENCH_FORCE_DIST=1 JL_EXCHANGE_STAGED=1 python3 bench.py $F > $O/staged.json 2> $O/staged.err || exit 1
python3 bench.py $F > $O/plain.json 2> $O/plain.err || exit 1
for f in bound worker staged plain; do python3 -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f',round(1000*d['ms_per_step'],2),'us/step;',d['config']['parallelism'][:90])"; done
