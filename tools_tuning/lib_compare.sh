# steady-state loop (group_loop.py 8 x 4) and the 20-window region for each alternative build of the library
for lib in "" $(ls minorseq_amd/libalt_*.so 2>/dev/null); do
  echo "== lib ${lib:-default}"
  JL_LIB=$lib python tools_tuning/group_loop.py 100000 3000 8 4 2>&1 | tail -1
  JL_LIB=$lib python tools_tuning/short_region.py 8,8,4 2>&1 | tail -2
done
