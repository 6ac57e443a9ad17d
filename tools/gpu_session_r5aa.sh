#!/bin/bash
# round 5, session aa: grid of build_keys_kernel (every block flushes its non-zero digit counters with global atomics)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5aa}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
( for n in 1048576 4194304; do
    for g in 128 256 512 1024 2048; do
      echo "== partitions $n LA_KEYS_GRID=$g"; LA_KEYS_GRID=$g timeout 200 python tools/cfg5_probe.py --partitions $n --reps 10 2>&1 | grep -E "^default"
    done
  done ) 2>&1 | grep -v amdgpu.ids > $O/sweep.txt
tail -1 $O/build.txt; cat $O/sweep.txt
