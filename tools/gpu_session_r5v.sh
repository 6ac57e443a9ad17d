#!/bin/bash
# round 5, session v: under-filled tile launches get twice the lanes per topic (latency, not throughput) -- A/B
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5v}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round5_gpu.py -q -m gpu -x 2>&1 | tail -2 > $O/tests.txt
B="--steps 3000 --warmup 200 --no-cpu-baseline --no-sort-phase --no-configs --no-live-traffic"
( for w in cfg3; do
    echo "== $w widened"; timeout 200 python bench.py --workload $w $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
    echo "== $w narrow";  LA_NO_TILE_WIDEN=1 timeout 200 python bench.py --workload $w $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
  for tpc in "200 256 32" "1000 64 8" "3000 64 8" "500 1024 64" "100 512 16" "2000 128 4"; do
    set -- $tpc
    echo "== custom $1 x $2 x $3: widened / narrow"
    timeout 200 python bench.py --workload custom --topics $1 --partitions $2 --consumers $3 --dist zipf $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    LA_NO_TILE_WIDEN=1 timeout 200 python bench.py --workload custom --topics $1 --partitions $2 --consumers $3 --dist zipf $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done ) > $O/ab.txt 2>&1
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$R/kafka_lag_based_assignor_amd -o /tmp/latency_c
( echo "== widened"; timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline; echo "== narrow"; LA_NO_TILE_WIDEN=1 timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline ) > $O/latency_c.txt 2>&1
cat $O/tests.txt; tail -1 $O/build.txt; grep -v amdgpu $O/ab.txt; cut -c1-200 $O/latency_c.txt
