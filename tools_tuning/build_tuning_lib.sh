# builds the library with -DJL_TUNING (probes, phase skipping) and the extra flags $1 into tools_tuning/lib_exp/${2:-libjuliet_hip.so};
# JL_LIB points tools at it
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/build/jl_tuning_build
rm -rf $B; mkdir -p $B; cp -r $R/minorseq_amd/csrc $B/; mkdir -p $B/include $B/tools_tuning/lib_exp
cp $R/include/juliet_hip.h $B/include/
mkdir -p $B/csrc_root/minorseq_amd; mv $B/csrc $B/csrc_root/minorseq_amd/csrc; cp -r $B/include $B/csrc_root/
make -s -j8 -C $B/csrc_root/minorseq_amd/csrc clean >/dev/null 2>&1 || true
make -s -j8 -C $B/csrc_root/minorseq_amd/csrc EXTRA="-DJL_TUNING $1"
mkdir -p $R/tools_tuning/lib_exp
cp $B/csrc_root/minorseq_amd/libjuliet_hip.so $R/tools_tuning/lib_exp/${2:-libjuliet_hip.so}
rm -rf $B
echo built $R/tools_tuning/lib_exp/${2:-libjuliet_hip.so}
