#!/bin/bash
# Round 4, first GPU session: the new entry points under test, the default bench line, the N > 1 bench legs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r4a_smoke.txt 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x 2>&1 | tail -40 > gpurun_out/r4a_tests_new.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "group or block or cfg2 or large_radix or frozen" 2>&1 | tail -15 > gpurun_out/r4a_tests_parity.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
timeout 1500 python -m pytest tests/test_bench_multirank_gpu.py -q -x 2>&1 | tail -30 > gpurun_out/r4a_tests_multirank.txt
tail -3 gpurun_out/r4a_tests_new.txt gpurun_out/r4a_tests_parity.txt gpurun_out/r4a_tests_multirank.txt
head -c 600 gpurun_out/r4a_bench.json
