#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4j_build.txt 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_gpu_parity.py -q -x -k "keys_first or group or small or zero or large_radix or phase" 2>&1 | tail -8 > $O/r4j_tests.txt
timeout 300 python tools/latency_probe.py > $O/r4j_latency.txt 2>&1
for rep in 1 2; do
  timeout 200 python bench.py --phase sort --steps 5 --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('keys first: sort %.4f ms keys %.4f ms ids %.4f ms frac %.4f keys_first %s sorted_ok %s' % (r['kernel_ms'], r['keys_ms'], r['ids_ms'], r['frac'], r.get('keys_first'), r['sorted_ok']))" >> $O/r4j_sort.txt 2>&1
done
tail -n 3 $O/r4j_tests.txt; cat $O/r4j_sort.txt $O/r4j_latency.txt
