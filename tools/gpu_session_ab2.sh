#!/bin/bash
# same-box A/B of two builds on block-path shapes: tools/_lab/a.so, tools/_lab/b.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
SH="1,10000,128 200,8000,16 1000,2000,100 20000,100,65 64,8192,2048"
for rep in 1 2; do for v in a b; do
  echo "== $v"; LA_LIB_PATH=tools/_lab/$v.so python tools/block_probe.py $SH 2>&1 | grep -v amdgpu
done; done > gpurun_out/ab/block.txt
cat gpurun_out/ab/block.txt
LA_LIB_PATH=tools/_lab/b.so python -m pytest tests -q -m gpu -k "block" 2>&1 | grep -E "passed|failed" | tail -2
