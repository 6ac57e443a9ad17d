#!/bin/bash
# round 4, session v (= r, the last commit of the round): the round's last kernels -- build + smoke, the full GPU suite, stress, the large-path probes, the driver's bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4v
export TMPDIR=/tmp
O=gpurun_out/r4v
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 > $O/tests.txt
timeout 600 python tools/stress_gpu.py 60 60 40 120 20 20 2>&1 | grep -v amdgpu.ids | tail -4 > $O/stress.txt
timeout 300 python tools/large_many_probe.py 2>&1 | grep -v amdgpu.ids > $O/large_many.txt
timeout 300 python tools/large_probe.py 2>&1 | grep -v amdgpu.ids > $O/large_probe.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cat $O/tests.txt $O/stress.txt $O/large_many.txt $O/large_probe.txt; tail -1 $O/build.txt
