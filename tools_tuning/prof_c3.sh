#!/bin/bash
# kernel durations of the configs[3] session step: rocprofv3 kernel trace of tools_tuning/config3_sharded_stages.py.
# usage (through gpurun, from the repo root): bash tools_tuning/prof_c3.sh [tag]   (JL_LIB selects an experimental build)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-c3}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/tools_tuning/config3_sharded_stages.py > $R/gpurun_out/prof_$TAG.log 2>&1
tail -1 $R/gpurun_out/prof_$TAG.log
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY
