#!/bin/bash
# round 5, session h: chunk-parallel placement of the one-workgroup grouping -- tests around grouping and small calls, C latency
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r5h}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu 2>&1 | tail -6 > $O/tests_new.txt
timeout 1500 python -m pytest tests -q -m gpu -x -k "group or small or zero_copy or reference or plugin or host or latency or member" 2>&1 | tail -4 > $O/tests_group.txt
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/latency_c
timeout 120 /tmp/latency_c > $O/latency_c.txt 2>&1
timeout 200 python tools/group_probe.py > $O/group_probe.txt 2>&1
cat $O/tests_new.txt $O/tests_group.txt; tail -1 $O/build.txt; cat $O/latency_c.txt; grep -v amdgpu $O/group_probe.txt | head -30
