# Whole builds (tools_tuning/ingest_time.py: host clock per build, 40 builds) under libraries that differ in -D flags, alternating,
# one stream and two, with and without qualities.   usage: build_ab.sh "<flags A>" "<flags B>"   (on the GPU box)
R=${GRAFT_REPO_ROOT:-.}
cd $R
bash tools_tuning/build_tuning_lib.sh "$1" libjuliet_a.so > /dev/null 2>&1 || { echo "build A failed"; exit 1; }
bash tools_tuning/build_tuning_lib.sh "$2" libjuliet_b.so > /dev/null 2>&1 || { echo "build B failed"; exit 1; }
for q in 20 0; do
  for one in 1 0; do
    for rep in 1 2 3; do
      for v in a b; do
        printf "min_qv %2d one_stream %d  %s: " $q $one $v
        JL_ING_ONE_STREAM=$one JL_LIB=$R/tools_tuning/lib_exp/libjuliet_$v.so python3 tools_tuning/ingest_time.py 100000 3000 40 $q 2>&1 | grep "per build" | sed 's/^.*builds: //'
      done
    done
  done
done
