#!/bin/bash
# Block path (one workgroup per topic) on batches of mid-size topics: bench lines + kernel stats.
# Usage: tools/gpu_session_block.sh TAG
TAG=${1:-r01_block}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for cfg in "1000 2000 100" "5000 200 100" "200 8000 16" "2000 1000 500" "20000 100 65"; do
  set -- $cfg
  timeout 300 python bench.py --topics $1 --partitions $2 --consumers $3 --steps 50 --warmup 10 > $O/bench_$1x$2x$3.json 2> $O/bench_$1x$2x$3.err
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --topics 1000 --partitions 2000 --consumers 100 --steps 50 --warmup 10 --no-cpu-baseline > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_8000x16 -- python $R/bench.py --topics 200 --partitions 8000 --consumers 16 --steps 50 --warmup 10 --no-cpu-baseline > $O/stats_8000x16.log 2>&1
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read()); print(d['config']['workload'][:60], '%.3e'%d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity'])"; done
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-150
find $O/stats_8000x16 -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-150
