#!/bin/bash
# round 5, session w: up to how many wavefronts does widening an under-filled tile launch pay?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5w}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
B="--steps 3000 --warmup 200 --no-cpu-baseline --no-sort-phase --no-configs --no-live-traffic"
( for tpc in "2000 256 32" "4000 256 32" "8000 256 32" "3000 64 8" "6000 64 8" "12000 64 8" "25000 64 8" "1500 1024 64" "4000 100 5"; do
    set -- $tpc
    echo "== $1 x $2 x $3: waves 0 (narrow) 1024 2048 4096 8192"
    for wv in 0 1024 2048 4096 8192; do
      LA_TILE_WIDEN_WAVES=$wv timeout 200 python bench.py --workload custom --topics $1 --partitions $2 --consumers $3 --dist zipf $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], end=' ')"
    done; echo
  done ) > $O/ab.txt 2>&1
tail -1 $O/build.txt; grep -v amdgpu $O/ab.txt
