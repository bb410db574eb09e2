cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
bash tools_tuning/build_tuning_lib.sh "-DJL_INGEST_NT_QV=1" > /dev/null 2>&1
export JL_ING_ONE_STREAM=1
{ echo "with qualities (min_qv 20), a build with ONE tile a workgroup (-DJL_INGEST_NT_QV=1: the store probes live in the one-tile store path):"
  MIN_QV=20 SKIPS="0 4 128 256 2048" bash tools_tuning/skip_ingest.sh; } > gpurun_out/r06/j_phases_off_qv_one_tile.txt 2>&1
cat gpurun_out/r06/j_phases_off_qv_one_tile.txt
bash tools_tuning/build_tuning_lib.sh > /dev/null 2>&1
