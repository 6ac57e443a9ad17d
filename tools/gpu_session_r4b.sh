#!/bin/bash
# Round 4, second GPU session: large topics side by side, > 8 192 consumers, why the sort phase read 3.3 ms, bench again.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build()" > $O/r4b_build.txt 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "large or consumers or huge" 2>&1 | tail -30 > $O/r4b_tests_new.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large or cfg5 or cfg2 or mixed or group" 2>&1 | tail -15 > $O/r4b_tests_parity.txt
timeout 600 python tools/large_many_probe.py > $O/r4b_large_many.txt 2>&1
# the sort phase: this tree against the round-3 tree, fresh processes, alternating
for rep in 1 2; do
  (cd tools/_lab/r3tree && timeout 200 python bench.py --phase sort --steps 5 --no-live-traffic) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('r3 tree : sort %.4f ms frac %.4f passes %d+%d' % (r['kernel_ms'], r['frac'], r['id_passes'], r['key_passes']))" >> $O/r4b_sort_ab.txt 2>&1
  timeout 200 python bench.py --phase sort --steps 5 --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('this tree: sort %.4f ms frac %.4f passes %d+%d' % (r['kernel_ms'], r['frac'], r['id_passes'], r['key_passes']))" >> $O/r4b_sort_ab.txt 2>&1
done
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4b_prof_sort -- python bench.py --phase sort --steps 3 --no-live-traffic > $O/r4b_prof_sort.log 2>&1
find $O/r4b_prof_sort -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r4b_sort_kernel_stats.csv
rm -rf $O/r4b_prof_sort
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4b_bench.json 2> $O/r4b_bench.err
tail -n 3 $O/r4b_tests_new.txt $O/r4b_tests_parity.txt
cat $O/r4b_large_many.txt $O/r4b_sort_ab.txt
head -c 400 $O/r4b_bench.json
