#!/bin/bash
# round 5, session i: full GPU suite at the state after wire-out / grouping / tree bins, latency probes (Python and C), stress
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r5i}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -8 > $O/tests.txt
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/latency.txt
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/latency_c
timeout 120 /tmp/latency_c > $O/latency_c.txt 2>&1
timeout 600 python tools/stress_gpu.py 40 40 40 60 20 20 2>&1 | grep -v amdgpu.ids | tail -4 > $O/stress.txt
cat $O/tests.txt; tail -1 $O/build.txt; cut -c1-330 $O/latency.txt; cat $O/latency_c.txt $O/stress.txt
