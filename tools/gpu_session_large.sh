#!/bin/bash
# Large-topic path (device radix sort) on a working set past the 256 MiB Infinity Cache: kernel stats + PMC traffic.
# Usage: tools/gpu_session_large.sh TAG [PARTITIONS]
TAG=${1:-r01_large}; P=${2:-33554432}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/large_probe.py --partitions $P --consumers 0 --launches 8 > $O/stats.log 2>&1
python3 $R/tools/trace_timeline.py $O/stats > $O/timeline.txt 2>&1
PROBE="python $R/tools/pmc_probe.py --topics 0 --large-partitions $P --large-consumers 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PROBE > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PROBE > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_parse.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.json 2> $O/pmc_parse.err
# cfg5 itself under the kernel trace: the one-workgroup greedy with sample-sorted rounds (+ the index -> rank pass)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -- python $R/tools/large_probe.py --partitions 1048576 --consumers 8192 --launches 8 --dist pareto > $O/stats_cfg5.log 2>&1
cd $R
for cfg in "1048576 8192" "10000 128"; do set -- $cfg; python tools/large_probe.py --partitions $1 --consumers $2 --launches 5 --dist pareto --check 2>&1 | grep "large topic\|bit-exact" >> $O/configs.txt; done
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
tail -3 $O/timeline.txt; cat $O/configs.txt
python3 - <<PY
import json
d=json.load(open("$O/pmc_summary.json"))
print(d["calibration"])
for k,e in d["kernels"].items():
    if "la::" in k: print(k.split("::")[-1][:40], {c:round(v) for c,v in e.items() if "bytes" in c or c=="dispatches" or c=="avg_ns_under_pmc"})
PY
