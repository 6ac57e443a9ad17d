#!/bin/bash
# Same-box A/B of two builds of the tile kernel: timing (bench, 1 000 steps) + the LDS / VALU counters.
# Usage: tools/pmc_tile.sh TAG LIB_A LIB_B      (library paths; "default" = the in-tree build)
TAG=$1; A=$2; B=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
for v in A B; do
  lib=$A; [ $v = B ] && lib=$B
  [ "$lib" = default ] && unset LA_LIB_PATH || export LA_LIB_PATH=$lib
  cd $R
  for rep in 1 2; do
    python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-sort-phase 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v rep$rep kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'value %.4g' % d['value'])" | tee -a $O/timing.txt
  done
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$v -- python $R/tools/pmc_probe.py --launches 5 > $O/pmc_$v.log 2>&1
  cd $R
  python tools/pmc_parse.py $O/pmc_$v > $O/pmc_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/pmc_$v.json"))
for k,e in d["kernels"].items():
    if "packed" in k:
        print("$v", k.split("::")[-1][:50], {c: round(x) for c,x in e.items()})
PY
done
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
