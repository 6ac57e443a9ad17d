#!/bin/bash
# Same-box A/B of builds of liblagassign.so: bench.py alternating over the libraries, three times each (one gpurun call = one box;
# only rows of the same call compare).  Libraries: "default" = the in-tree build, anything else a path relative to the repository
# (tools/build_lab_lib.sh builds them under tools/_lab/).
# Usage: tools/ab_bench.sh TAG assign|sort "extra bench args" LIB [LIB ...]
#   tools/ab_bench.sh r03_tile_full assign "" default tools/_lab/liblagassign_base.so
#   TESTS=1 ... also runs the tile-path parity tests on the default build first
TAG=$1; WHAT=$2; ARGS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/$TAG; mkdir -p $O
if [ -n "$TESTS" ]; then
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite_gpu.py -x -q -m gpu -k "not large and not cfg5 and not fuzz_large" > $O/pytest_tile.log 2>&1
  echo "pytest rc=$?" | tee $O/summary.txt; tail -2 $O/pytest_tile.log | tee -a $O/summary.txt
fi
for rep in 1 2 3; do
  for lib in "$@"; do
    [ "$lib" = default ] && unset LA_LIB_PATH || export LA_LIB_PATH=$R/$lib
    if [ "$WHAT" = sort ]; then
      timeout 200 python bench.py --phase sort --steps 10 --no-live-traffic $ARGS > $O/b.json 2> $O/b.err
    else
      timeout 120 python bench.py --steps 1000 --no-cpu-baseline --no-sort-phase --no-live-traffic $ARGS > $O/b.json 2> $O/b.err
    fi
    python3 - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
    if "$WHAT" == "sort": print("$lib rep $rep: sort %.4f ms frac %.4f sorted_ok %s" % (r["kernel_ms"], r["frac"], r["sorted_ok"]))
    else: print("$lib rep $rep: ms_per_step %.4f kernel_ms %.4f frac %.4f cold %.4f" % (d["ms_per_step"], r["kernel_ms"], r["frac"], d["cold_call_ms"]))
except Exception as e:
    print("$lib rep $rep failed", e, open("$O/b.err").read()[-300:])
PY
  done
done
