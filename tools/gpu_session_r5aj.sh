#!/bin/bash
# round 5, session aj: the full GPU suite on the round's last commit
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5aj}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 280 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|error|Error" | tail -6 > $O/tests.txt
cat $O/tests.txt; tail -1 $O/build.txt
