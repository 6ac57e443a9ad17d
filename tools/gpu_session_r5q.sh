#!/bin/bash
# round 5, session q: the staged form's packing and unpacking handed out to the context's parked threads
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5q}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
export LAT_ROWS=1000x50x5,1000x100x8,1000x256x32,2000x256x32
( echo "== default (6 MB), serial copies"; LA_NO_PARALLEL_COPY=1 timeout 300 python tools/latency_probe.py
  echo "== default (6 MB)"; timeout 300 python tools/latency_probe.py
  for z in 12582912 25165824; do
    echo "== LA_ZERO_COPY_BYTES=LA_SMALL_BYTES=$z"
    LA_SMALL_BYTES=$z LA_ZERO_COPY_BYTES=$z timeout 300 python tools/latency_probe.py
  done ) 2>&1 | grep -v amdgpu.ids > $O/latency.txt
tail -1 $O/build.txt; cut -c1-250 $O/latency.txt
