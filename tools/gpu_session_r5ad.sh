#!/bin/bash
# round 5, session ad: soak of the zero-copy host calls (every result checked)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5ad}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
( timeout 200 python tools/soak_small_calls.py 90
  LA_NO_TILE_WIDEN=1 timeout 100 python tools/soak_small_calls.py 30 ) 2>&1 | grep -v amdgpu.ids | tail -8 > $O/soak.txt
tail -1 $O/build.txt; cat $O/soak.txt
