#!/bin/bash
# round 5, session s: the mid-size grouping ends the zero-copy call itself (one launch less)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5s}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py tests/test_multi_device_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8 > $O/tests.txt
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$R/kafka_lag_based_assignor_amd -o /tmp/latency_c && timeout 120 /tmp/latency_c oracle/liblagoracle.so > $O/latency_c.txt 2>&1
for i in 1 2 3; do timeout 200 python tools/stress_gpu.py 10 0 0 0 0 20 2>&1 | tail -1; done > $O/stress.txt
cat $O/tests.txt; tail -1 $O/build.txt; grep -A1 "10000 part" $O/latency_c.txt; cat $O/stress.txt
