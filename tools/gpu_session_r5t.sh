#!/bin/bash
# round 5, session t: randomized parity sweep with fresh seeds, incl. the round-5 sections (32-bit-key greedy, host calls of every size)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5t}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
( STRESS_SEED0=20000 timeout 1100 python tools/stress_gpu.py 300 100 300 60 40 40 1500
  date ) 2>&1 | grep -v amdgpu.ids | tail -12 > $O/stress.txt
tail -1 $O/build.txt; cat $O/stress.txt
