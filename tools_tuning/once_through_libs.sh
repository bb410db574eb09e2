# the once_through leg of bench.py with tuning builds of the library (LDS padding of the planes kernel = fewer of its workgroups per
# CU, so that the other windows' kernels run beside it)
for L in "$@"; do
  JL_LIB=$GRAFT_REPO_ROOT/tools_tuning/lib_exp/$L python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-config3 --no-config4 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);o=d['once_through'];print('$L once_through',round(1000*o['ms_per_step'],1),'us/step; ingest alone',round(1000*o['ingest_ms'],1))"
done
