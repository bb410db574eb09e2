set -e
cd $GRAFT_REPO_ROOT
python -c "import torch" 
for rep in 1 2 3; do
for gi in "8 4" "5 4" "10 2" "20 1" "4 4" "7 3" "10 4"; do
  set -- $gi
  echo "== G=$1 I=$2 rep=$rep" >> gpurun_out/sweep20.log
  python bench.py --steps 20 --warmup 5 --group $1 --inflight $2 --no-cpu-baseline >> gpurun_out/sweep20.log 2>gpurun_out/sweep20.err
done
done
