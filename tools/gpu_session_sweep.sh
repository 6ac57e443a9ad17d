#!/bin/bash
# Same-box A/B of the single-kernel radix passes: this build against another liblagassign.so (LA_LIB_PATH).
# Usage: tools/gpu_session_sweep.sh TAG [OTHER_LIB]      (OTHER_LIB default tools/_lab/liblagassign_base.so)
TAG=${1:-sweep}; OTHER=${2:-tools/_lab/liblagassign_base.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large or radix or cfg5 or group_by_member or mixed" > $O/pytest_large.log 2>&1
echo "pytest large rc=$?" | tee $O/summary.txt
tail -3 $O/pytest_large.log | tee -a $O/summary.txt
for rep in 1 2; do
  for lib in default $OTHER; do
    [ "$lib" = default ] && unset LA_LIB_PATH || export LA_LIB_PATH=$R/$lib
    timeout 300 python bench.py --phase sort --steps 10 --no-live-traffic > $O/sort_${rep}_$(basename $lib .so).json 2> $O/sort_${rep}_$(basename $lib .so).err
    python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/sort_${rep}_$(basename $lib .so).json").read().strip().splitlines()[-1])["roofline"]
    print("$lib rep $rep: sort %.4f ms frac %.4f sorted_ok %s" % (d["kernel_ms"], d["frac"], d["sorted_ok"]))
except Exception as e:
    print("$lib rep $rep: failed", e)
PY
  done
done
for lib in default $OTHER; do
  [ "$lib" = default ] && unset LA_LIB_PATH || export LA_LIB_PATH=$R/$lib
  for cfg in "1048576 8192" "4194304 0" "8388608 0"; do set -- $cfg
    echo "$lib: $(timeout 300 python tools/large_probe.py --partitions $1 --consumers $2 --launches 5 --dist pareto --check 2>&1 | grep 'large topic\|bit-exact' | tr '\n' ' ')" | tee -a $O/summary.txt
  done
done
unset LA_LIB_PATH
