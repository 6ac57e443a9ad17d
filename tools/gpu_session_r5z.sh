#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5z}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -q -m gpu -k "widened or narrow or mid_size or staged" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt; tail -1 $O/build.txt
