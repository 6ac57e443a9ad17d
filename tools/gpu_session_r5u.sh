#!/bin/bash
# round 5, session u: tile size of the one-kernel radix passes at mid sizes (cfg5's 1 M partitions run 128 tiles of 8 192 on 256 CUs)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5u}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
( for n in 262144 1048576 2097152 4194304; do
    for w in 256 512 1024; do
      echo "== partitions $n LA_SWEEP_THREADS=$w"; LA_SWEEP_THREADS=$w timeout 200 python tools/cfg5_probe.py --partitions $n --reps 10 2>&1 | grep -E "^default"
    done
  done ) 2>&1 | grep -v amdgpu.ids > $O/sweep.txt
tail -1 $O/build.txt; cat $O/sweep.txt
