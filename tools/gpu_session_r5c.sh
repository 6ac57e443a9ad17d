#!/bin/bash
# round 5, session c: the 32-bit-key greedy of the block path (A/B against LA_BLOCK_KEY32=0), the LDS-staged grouping, C-level latency
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r5c}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_new.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -8 > $O/tests.txt
SH="1,10000,128 1,8000,256 1,16000,200 1,3000,100 600,300,128 64,8192,2048 200,8000,16"
for m in 1 0 1 0; do
  echo "== LA_BLOCK_KEY32=$m" >> $O/block_ab.txt
  LA_BLOCK_KEY32=$m timeout 300 python tools/block_probe.py $SH 2>&1 | grep -v amdgpu | cut -c1-100 >> $O/block_ab.txt
done
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/latency_c
timeout 120 /tmp/latency_c > $O/latency_c.txt 2>&1
LA_NO_FUSED_TAIL=1 timeout 120 /tmp/latency_c > $O/latency_c_nofuse.txt 2>&1
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/latency.txt
cat $O/tests_new.txt $O/tests.txt; tail -1 $O/build.txt; cat $O/block_ab.txt; cat $O/latency_c.txt; echo nofuse; cat $O/latency_c_nofuse.txt; cut -c1-420 $O/latency.txt
