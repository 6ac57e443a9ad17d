#!/bin/bash
# round 5, session ae: the fused end of a small call (256 threads of the tile kernel's last workgroup build the lists) against the
# separate one-workgroup grouping launch (1 024 threads), by size
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5ae}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$R/kafka_lag_based_assignor_amd -o /tmp/latency_c
( for i in 1 2; do
  echo "== fused"; timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline
  echo "== LA_NO_FUSED_TAIL=1"; LA_NO_FUSED_TAIL=1 timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline
  done ) > $O/latency_c.txt 2>&1
tail -1 $O/build.txt; cut -c1-175 $O/latency_c.txt
