#!/bin/bash
# round 5, session x: widen by the cost model with the bins' network growing with L (kept so far) or not at all (what the kernel does)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5x}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
B="--steps 3000 --warmup 200 --no-cpu-baseline --no-sort-phase --no-configs --no-live-traffic"
( for tpc in "4000 100 5" "3000 64 8" "2000 128 4" "100 512 16" "200 1000 3" "1000 50 5" "40 50 5" "500 200 20" "6000 64 8"; do
    set -- $tpc
    echo "== $1 x $2 x $3: model / full"
    timeout 200 python bench.py --workload custom --topics $1 --partitions $2 --consumers $3 --dist zipf $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    LA_TILE_WIDEN_FULL=1 timeout 200 python bench.py --workload custom --topics $1 --partitions $2 --consumers $3 --dist zipf $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done ) > $O/ab.txt 2>&1
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$R/kafka_lag_based_assignor_amd -o /tmp/latency_c
( echo "== model"; timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline; echo "== full"; LA_TILE_WIDEN_FULL=1 timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline ) > $O/latency_c.txt 2>&1
tail -1 $O/build.txt; cat $O/ab.txt; cut -c1-200 $O/latency_c.txt
