#!/bin/bash
# round 4, session q: the full GPU suite, the driver's bench command and a kernel trace of cfg5 after the moved-bins sort
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4q
export TMPDIR=/tmp
O=gpurun_out/r4q
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 > $O/tests.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg5 -o cfg5 -- python $OLDPWD/tools/cfg5_probe.py --reps 10 > /dev/null 2>&1; f=$(find /tmp/prof_cfg5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OLDPWD/$O/cfg5_kernel_stats.csv)
cat $O/tests.txt; tail -2 $O/build.txt; head -c 1500 $O/bench.json; echo; head -8 $O/cfg5_kernel_stats.csv
