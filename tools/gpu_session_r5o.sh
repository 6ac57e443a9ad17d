#!/bin/bash
# round 5, session o: how far up does the zero-copy form (pack into mapped staging, one launch, spin) beat lanes / mapped?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5o}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
export LAT_ROWS=300x100x8,1000x50x5,1000x100x8,1000x256x32
( for z in 2097152 4194304 8388608 16777216; do
    echo "== LA_ZERO_COPY_BYTES=LA_SMALL_BYTES=$z"
    LA_SMALL_BYTES=$z LA_ZERO_COPY_BYTES=$z timeout 300 python tools/latency_probe.py
  done ) 2>&1 | grep -v amdgpu.ids > $O/latency.txt
tail -1 $O/build.txt; cut -c1-250 $O/latency.txt
