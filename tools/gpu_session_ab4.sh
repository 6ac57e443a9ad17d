#!/bin/bash
# block path: the workgroup's sort by digits (LA_BLOCK_RADIX=0 never / 1 the three largest classes / 2 every class), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
SH="1,10000,128 1,16000,200 1,8000,256 200,8000,16 200,5000,16 64,8192,2048 300,4000,100 1000,2000,100 1000,1100,100 2000,1000,500 5000,200,100 20000,100,65 20000,300,10"
for m in 1 2; do
  echo "== LA_BLOCK_RADIX=$m tests:"; LA_BLOCK_RADIX=$m python -m pytest tests -q -m gpu -k "block or reference or fuzz or ragged" 2>&1 | grep -E "passed|failed" | tail -2
done > gpurun_out/ab/radix.txt
for rep in 1 2; do for m in 0 1 2; do
  echo "== LA_BLOCK_RADIX=$m"; LA_BLOCK_RADIX=$m python tools/block_probe.py $SH 2>&1 | grep -v amdgpu | cut -c1-60
done; done >> gpurun_out/ab/radix.txt
cat gpurun_out/ab/radix.txt
