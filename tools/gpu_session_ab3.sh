#!/bin/bash
# large path, few consumers: bins per lane of the one-workgroup greedy (lab build tools/_lab/r.so reads LA_ROUNDS_EC)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
for c in 1024 512 256 128; do for ec in 0 2 4 8; do
  echo -n "C=$c LA_ROUNDS_EC=$ec: "; LA_ROUNDS_EC=$ec LA_LIB_PATH=tools/_lab/r.so python tools/large_probe.py --partitions 2097152 --consumers $c --launches 2 --check 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
done; done > gpurun_out/ab/rounds_ec.txt
cat gpurun_out/ab/rounds_ec.txt
