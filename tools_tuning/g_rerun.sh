cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for i in 1 2; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2951$i JL_BENCH_FORCE_DIST=1 python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-once-through > gpurun_out/r06/g2_bound_$i.json 2>> gpurun_out/r06/g2.err
python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-once-through --no-end-to-end > gpurun_out/r06/g2_plain_$i.json 2>> gpurun_out/r06/g2.err
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2952$i JL_BENCH_FORCE_DIST=1 JL_NO_FOLD_CALL=1 python3 bench.py --steps 4000 --warmup 64 --no-cpu-baseline --no-config3 --no-once-through > gpurun_out/r06/g2_bound_nofold_$i.json 2>> gpurun_out/r06/g2.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/g2_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"]*1000,2))
    except Exception as e: print(f, "failed", e)
PY
