#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4l_build.txt 2>&1
timeout 1500 python tools/stress_gpu.py 20 10 20 10 10 60 > $O/r4l_stress.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "group" 2>&1 | tail -3 >> $O/r4l_stress.txt
grep -v amdgpu.ids $O/r4l_stress.txt | tail -20
