#!/bin/bash
# same-box A/B of two builds of liblagassign.so on cfg5: tools/_lab/a.so, tools/_lab/b.so (tools/build_lab_lib.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
for rep in 1 2 3; do for v in ${AB_VARIANTS:-a b}; do
  echo -n "$v: "; LA_LIB_PATH=tools/_lab/$v.so python tools/cfg5_probe.py --reps 20 2>&1 | grep -E "^default"
done; done > gpurun_out/ab/cfg5.txt
cat gpurun_out/ab/cfg5.txt
