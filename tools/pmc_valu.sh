#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for a in "$@"; do
  rm -rf /tmp/pv_$a
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pv_$a -- $R/tools/_lab/lab$a > /dev/null 2>&1
  python3 $R/tools/pmc_parse.py /tmp/pv_$a | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,e in d['kernels'].items():
    if 'packed' in k: print('lab$a', k, 'VALU/wave %.0f SALU/wave %.0f LDS/wave %.0f  ns %.0f  kcycles %.0f' % (e['SQ_INSTS_VALU']/e['SQ_WAVES'], e['SQ_INSTS_SALU']/e['SQ_WAVES'], e['SQ_INSTS_LDS']/e['SQ_WAVES'], e.get('avg_ns_under_pmc',0), e.get('GRBM_GUI_ACTIVE',0)/8e3))
"
done
