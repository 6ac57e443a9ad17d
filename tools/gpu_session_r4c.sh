#!/bin/bash
# Round 4, third GPU session: linear small grouping, zero-copy small calls, sparse list uploaded once; latency A/B; bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4c_build.txt 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x 2>&1 | tail -30 > $O/r4c_tests_new.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_device_gpu.py tests/test_reference_suite_gpu.py -q -x -k "group or small or grouped or reference or plugin or assign" 2>&1 | tail -15 > $O/r4c_tests_parity.txt
timeout 300 python tools/latency_probe.py > $O/r4c_latency.txt 2>&1
LA_ZERO_COPY_BYTES=0 timeout 300 python tools/latency_probe.py > $O/r4c_latency_onecopy.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4c_bench.json 2> $O/r4c_bench.err
tail -n 3 $O/r4c_tests_new.txt $O/r4c_tests_parity.txt
cat $O/r4c_latency.txt $O/r4c_latency_onecopy.txt
