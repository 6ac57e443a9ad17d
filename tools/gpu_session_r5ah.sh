#!/bin/bash
# round 5, session ah: block path, lag from offsets in two stages (fewer spills in the E = 16 kernel)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5ah}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py tests/test_round5_gpu.py -q -m gpu -k "block or key32 or sentinel or cfg2 or ragged_batch_all or sparse" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5 > $O/tests.txt
STRESS_SEED0=61000 timeout 300 python tools/stress_gpu.py 0 0 150 0 0 0 100 2>&1 | tail -1 > $O/stress.txt
LAG_BITS=40 timeout 200 python tools/block_probe.py 1,10000,128 1,16000,200 200,8000,16 1000,2000,100 2>&1 | grep "T=" > $O/probe.txt
cat $O/tests.txt $O/stress.txt; tail -1 $O/build.txt; cat $O/probe.txt
