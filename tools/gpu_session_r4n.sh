#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4n_build.txt 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "bounds" 2>&1 | tail -5 > $O/r4n_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite_gpu.py -q -x -k "tile or fuzz or device_entry or reference or packed" 2>&1 | tail -4 >> $O/r4n_tests.txt
for rep in 1 2 3; do
  for nb in 1 0; do
    if [ $nb = 1 ]; then export LA_BENCH_NO_BOUNDS=1; else unset LA_BENCH_NO_BOUNDS; fi
    timeout 200 python bench.py --steps 1000 --no-cpu-baseline --no-sort-phase --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('no_bounds=$nb: ms_per_step %.4f kernel_ms %.4f frac %.4f value %.4g' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))" >> $O/r4n_ab.txt 2>&1
  done
done
unset LA_BENCH_NO_BOUNDS
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4n_bench.json 2> $O/r4n_bench.err
cat $O/r4n_tests.txt | grep -v amdgpu; cat $O/r4n_ab.txt
