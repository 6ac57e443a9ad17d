#!/bin/bash
# round 5, session n: new tests; where the zero-copy form (kernels read the staging buffer over PCIe, one launch, spin on a flag)
# hands over to the one-copy form now that the tail is fused
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5n}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_gpu_parity.py -q -m gpu -x -k "mapped or hint or lags or small or one_output or refus" 2>&1 | tail -3 > $O/tests.txt
export LAT_ROWS=10x10x3,100x20x4,50x100x8,100x100x8,1000x16x4,300x100x8
( for z in default 262144 524288 1048576 2097152; do
    echo "== LA_ZERO_COPY_BYTES=$z"
    if [ $z = default ]; then timeout 300 python tools/latency_probe.py; else LA_ZERO_COPY_BYTES=$z timeout 300 python tools/latency_probe.py; fi
  done ) 2>&1 | grep -v amdgpu.ids > $O/latency.txt
cat $O/tests.txt; tail -1 $O/build.txt; cut -c1-250 $O/latency.txt
