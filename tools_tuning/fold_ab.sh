# A/B of the Fisher stage folded into the pileup launch (default) against the separate call launch (JL_NO_FOLD_CALL=1), alternating
# on ONE box: the driver's 20-step command and the steady state.   usage: fold_ab.sh [rounds]
R=${GRAFT_REPO_ROOT:-.}
N=${1:-3}
O=$R/gpurun_out/fold_ab
mkdir -p $O
for i in $(seq 1 $N); do
  for v in fold nofold; do
    if [ $v = nofold ]; then export JL_NO_FOLD_CALL=1; else unset JL_NO_FOLD_CALL; fi
    python3 $R/bench.py --steps 20 --warmup 5 --no-config3 --no-cpu-baseline --no-once-through --no-end-to-end > $O/${v}_20_$i.json 2>/dev/null
    python3 $R/bench.py --no-config3 --no-cpu-baseline --no-once-through --no-end-to-end > $O/${v}_long_$i.json 2>/dev/null
  done
done
unset JL_NO_FOLD_CALL
python3 - <<PY
import json, glob
for v in ("fold", "nofold"):
    for kind in ("20", "long"):
        rows = []
        for f in sorted(glob.glob("$O/%s_%s_*.json" % (v, kind))):
            try:
                d = json.loads(open(f).read().strip().splitlines()[-1])
                c = d["config"]
                rows.append((d["ms_per_step"], d["roofline"]["kernel_ms"], c["one_batch_latency_c_abi_ms"], c["many_positions_latency_c_abi_ms"]))
            except Exception as e:
                rows.append(("failed", str(e)))
        print(v, kind, " | ".join(" ".join(f"{x:.5f}" if isinstance(x, float) else str(x) for x in r) for r in rows))
PY
