#!/bin/bash
# A second liblagassign.so with another la_large.o (development flags, or an older source), for same-box A/B via LA_LIB_PATH.
# Usage: tools/build_lab_lib.sh OUT.so LA_LARGE_SOURCE [hipcc flags...]
OUT=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/kafka_lag_based_assignor_amd/csrc
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$C -I$R/include "$@" -c $SRC -o $T/la_large.o || exit 1
OBJS=$(ls $C/build/*.o | grep -v la_large.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $OUT $OBJS $T/la_large.o -ldl && rm -rf $T && echo built $OUT
