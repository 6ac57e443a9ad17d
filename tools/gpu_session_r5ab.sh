#!/bin/bash
# round 5, session ab: window of the look-back walk (predecessors polled at once) at sizes where every tile runs at the same time
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5ab}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
( for n in 262144 1048576 4194304; do
    echo "== partitions $n window 2 (product)"; timeout 200 python tools/cfg5_probe.py --partitions $n --reps 10 2>&1 | grep -E "^default"
    for w in 4 8 16; do
      echo "== partitions $n window $w"; LA_LIB_PATH=$R/tools/_lab/look$w.so timeout 200 python tools/cfg5_probe.py --partitions $n --reps 10 2>&1 | grep -E "^default"
    done
  done
  echo "== sort phase 33.5 M: window 2 / 8 / 16"
  for lib in "" $R/tools/_lab/look8.so $R/tools/_lab/look16.so; do
    LA_LIB_PATH=$lib timeout 300 python bench.py --phase sort --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('roofline',{}).get('kernel_ms'))"
  done ) 2>&1 | grep -v amdgpu.ids > $O/sweep.txt
tail -1 $O/build.txt; cat $O/sweep.txt
