#!/bin/bash
# A second liblagassign.so for same-box A/B via LA_LIB_PATH: some translation units rebuilt with other flags or from other
# sources, the rest taken from the in-tree build (csrc/build/*.o).
# Usage: tools/build_lab_lib.sh OUT.so "UNIT[=SOURCE] ..." [hipcc flags...]
#   tools/build_lab_lib.sh tools/_lab/a.so "la_large=/tmp/old_la_large.hip" -DLA_SWEEP_CLOCKS
#   tools/build_lab_lib.sh tools/_lab/b.so "la_wave_tile_l8 la_wave_tile_l16 la_wave_tile_l32 la_wave_tile_l64" -DLA_WPE=7
OUT=$1; UNITS=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/kafka_lag_based_assignor_amd/csrc
T=$(mktemp -d)
SKIP=""
for u in $UNITS; do
  name=${u%%=*}; src=$C/$name.hip; [ "$u" != "$name" ] && src=${u#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$C -I$R/include "$@" -c $src -o $T/$name.o || touch $T/failed ) &
  SKIP="$SKIP -e /$name.o"
done
wait
[ -e $T/failed ] && { echo "compile failed"; exit 1; }
OBJS=$(ls $C/build/*.o | grep -v $SKIP)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $OUT $OBJS $T/*.o -ldl && rm -rf $T && echo built $OUT
