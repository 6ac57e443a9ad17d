#!/bin/bash
# Round 4, profile session: the native packed all-gather test, then tools/gpu_session.sh (bench lines, rocprofv3 kernel stats of
# the default command, PMC passes), the large-path session (sort phase timeline + PMC), and a kernel trace of small calls.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4i_build.txt 2>&1
timeout 900 python -m pytest tests/test_multi_device_gpu.py -q -x -k "rccl or device_entry" 2>&1 | tail -8 > $O/r4i_tests_rccl.txt
SKIP_TESTS=1 bash tools/gpu_session.sh r04_e > $O/r4i_session.log 2>&1
bash tools/gpu_session_large.sh r04_large > $O/r4i_session_large.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/r4i_lat -- python $OLDPWD/tools/latency_probe.py > $OLDPWD/$O/r4i_lat.log 2>&1
cd $OLDPWD
find $O/r4i_lat -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r4i_latency_kernel_stats.csv
rm -rf $O/r4i_lat
tail -n 3 $O/r4i_tests_rccl.txt; tail -5 $O/r4i_session.log; tail -5 $O/r4i_session_large.log; du -sh $O
