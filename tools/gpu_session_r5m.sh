#!/bin/bash
# round 5, session m: mapped caller arrays between the zero-copy and the one-copy thresholds -- read in place or staged?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5m}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt

export LAT_ROWS=100x20x4,100x100x8,1000x16x4,1000x50x5,200x256x32
( echo "== default"; timeout 300 python tools/latency_probe.py
  echo "== LA_MAPPED_SMALL_GROUPED=1"; LA_MAPPED_SMALL_GROUPED=1 timeout 300 python tools/latency_probe.py ) 2>&1 | grep -v amdgpu.ids > $O/latency.txt
tail -1 $O/build.txt; cut -c1-60,200-600 $O/latency.txt
