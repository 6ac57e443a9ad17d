#!/bin/bash
# round 5, session g: LA_FLAG_WIRE_OUT, coalesced lists of the small rebalance -- new tests, bench multirank tests, C latency, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r5g}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu 2>&1 | tail -12 > $O/tests_new.txt
timeout 1500 python -m pytest tests/test_bench_multirank_gpu.py tests/test_multi_device_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/tests_multi.txt
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/latency_c
timeout 120 /tmp/latency_c > $O/latency_c.txt 2>&1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-sort-phase > $O/bench.json 2> $O/bench.err
cat $O/tests_new.txt $O/tests_multi.txt; tail -1 $O/build.txt; cat $O/latency_c.txt; tail -2 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"].get("wire_out"))
print(d["small_call"]["rows"][:3])
PY
