#!/bin/bash
# round 5, session af: the new policy for the fused end (lists up to 1 024 entries only) -- tests, latency at the C ABI and through Python
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5af}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6 > $O/tests.txt
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$R/kafka_lag_based_assignor_amd -o /tmp/latency_c
( echo "== default"; timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline
  echo "== LA_FUSED_TAIL=all"; LA_FUSED_TAIL=all timeout 120 /tmp/latency_c oracle/liblagoracle.so | grep pipeline ) > $O/latency_c.txt 2>&1
timeout 100 python tools/soak_small_calls.py 30 2>&1 | tail -1 > $O/soak.txt
cat $O/tests.txt; tail -1 $O/build.txt; cut -c1-175 $O/latency_c.txt; cat $O/soak.txt
