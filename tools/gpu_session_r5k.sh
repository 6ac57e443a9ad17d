#!/bin/bash
# round 5, session k: the two-launch grouping of mid-size rebalances -- tests, group probe, latency probes, block-path kernel stats
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5k}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu 2>&1 | tail -6 > $O/tests_new.txt
timeout 1500 python -m pytest tests -q -m gpu -x -k "group or member or reference or plugin or host or multi_device or shard" 2>&1 | tail -4 > $O/tests_group.txt
timeout 200 python tools/group_probe.py > $O/group_probe.txt 2>&1
LA_NO_MID_GROUP=1 timeout 200 python tools/group_probe.py >> $O/group_probe.txt 2>&1
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/latency_c
timeout 120 /tmp/latency_c oracle/liblagoracle.so > $O/latency_c.txt 2>&1
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/latency.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_block -- python $R/tools/block_probe.py 1,10000,128 200,8000,16 1000,2000,100 > $O/stats_block.log 2>&1
cd $R
find $O -name "*.db" -delete 2>/dev/null
cat $O/tests_new.txt $O/tests_group.txt; tail -1 $O/build.txt; grep -v amdgpu $O/group_probe.txt; cat $O/latency_c.txt; cut -c1-330 $O/latency.txt | tail -6
