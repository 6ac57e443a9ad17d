#!/bin/bash
# round 4, session o: the moved-bins sort of the greedy rounds (cfg5's chain)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4o
O=gpurun_out/r4o
python tools/cfg5_probe.py --reps 10 > $O/cfg5_probe.txt 2>&1
LA_LIB_PATH=tools/_lab/clocks.so python tools/cfg5_probe.py --reps 4 > $O/cfg5_clocks.txt 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -k "sort_only_the_bins_that_move or merge_ascending_runs or cfg5 or large" 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/tests.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/cfg5_probe.py --reps 10 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg5_kernel_stats.csv; rm -rf $O/stats
cat $O/tests.txt; grep -E "^default|cycles" $O/cfg5_probe.txt $O/cfg5_clocks.txt | head -4; head -6 $O/cfg5_kernel_stats.csv
