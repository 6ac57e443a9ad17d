#!/bin/bash
# round 5, session b: one launch per small rebalance, the parked worker pool; latency probe (pageable, pinned; chunk sizes of the lanes pipeline)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r5b}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_new.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -8 > $O/tests.txt
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/latency.txt
LA_NO_FUSED_TAIL=1 timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | head -6 > $O/latency_nofuse.txt
for cp in 32768 65536 131072; do
  echo "== LA_CHUNK_PARTITIONS=$cp" >> $O/latency_chunks.txt
  LA_CHUNK_PARTITIONS=$cp timeout 300 python tools/latency_probe.py 2>&1 | grep -E "lanes|streams" >> $O/latency_chunks.txt
done
cat $O/tests_new.txt $O/tests.txt; tail -1 $O/build.txt; cat $O/latency.txt $O/latency_nofuse.txt $O/latency_chunks.txt | cut -c1-400
