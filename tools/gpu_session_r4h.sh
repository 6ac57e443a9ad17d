#!/bin/bash
# Round 4: the mapped pipeline for pinned callers; full GPU suite; bench; mapped probe again.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4h_build.txt 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_multi_device_gpu.py -q -x -k "sparse or pinned or stream or small or zero" 2>&1 | tail -30 > $O/r4h_tests_new.txt
timeout 600 python tools/mapped_probe.py > $O/r4h_mapped.txt 2>&1
timeout 1800 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 > $O/r4h_tests_all.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4h_bench.json 2> $O/r4h_bench.err
timeout 600 python tools/stress_gpu.py 100 40 60 20 40 > $O/r4h_stress.txt 2>&1
tail -n 4 $O/r4h_tests_new.txt $O/r4h_tests_all.txt $O/r4h_stress.txt
cat $O/r4h_mapped.txt
