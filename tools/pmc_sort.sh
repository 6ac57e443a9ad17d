#!/bin/bash
# PMC passes over the large path's sort kernels (one 33.5 M-partition topic).  Usage: tools/pmc_sort.sh TAG [multi|single]
TAG=$1; FORM=${2:-multi}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
[ "$FORM" = multi ] && export LA_SORT_MULTIKERNEL=1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_CYCLES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum TCC_REQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/large_probe.py --partitions 33554432 --consumers 0 --launches 2 > $O/p$i.log 2>&1
done
cd $R
python tools/pmc_parse.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 > $O/summary.json 2> $O/parse.err
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
python - <<PY
import json
d=json.load(open("$O/summary.json"))
for k,e in d["kernels"].items():
    if "scatter" in k or "onesweep" in k or "count" in k:
        print(k)
        for c,v in sorted(e.items()): print("   %-32s %.4g" % (c,v))
PY
