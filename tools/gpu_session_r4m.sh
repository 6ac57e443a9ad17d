#!/bin/bash
# Round 4, closing session: build + smoke, the whole GPU suite, the driver's bench command, a short stress sweep.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4m_build.txt 2>&1
timeout 1800 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 > $O/r4m_tests_all.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4m_bench.json 2> $O/r4m_bench.err ) 2> $O/r4m_bench_time.txt
timeout 900 python tools/stress_gpu.py 40 20 40 20 20 40 > $O/r4m_stress.txt 2>&1
timeout 300 python tools/latency_probe.py > $O/r4m_latency.txt 2>&1
tail -n 3 $O/r4m_build.txt $O/r4m_tests_all.txt $O/r4m_stress.txt; cat $O/r4m_bench_time.txt; head -c 300 $O/r4m_bench.json
