#!/bin/bash
# Round 4, sixth GPU session: keys-first sorts with the streaming tie scan; sort A/B; bench; kernel stats of the sort.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r4f_build.txt 2>&1
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "keys_first or large or consumers" 2>&1 | tail -30 > $O/r4f_tests_new.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "large or cfg5 or group" 2>&1 | tail -8 > $O/r4f_tests_parity.txt
for rep in 1 2; do
  LA_SORT_KEYS_FIRST=0 timeout 200 python bench.py --phase sort --steps 5 --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ids first : sort %.4f ms frac %.4f passes %d+%d keys_first %s' % (r['kernel_ms'], r['frac'], r['id_passes'], r['key_passes'], r.get('keys_first')))" >> $O/r4f_sort_ab.txt 2>&1
  timeout 200 python bench.py --phase sort --steps 5 --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('keys first: sort %.4f ms frac %.4f passes %d+%d keys_first %s sorted_ok %s' % (r['kernel_ms'], r['frac'], r['id_passes'], r['key_passes'], r.get('keys_first'), r['sorted_ok']))" >> $O/r4f_sort_ab.txt 2>&1
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4f_bench.json 2> $O/r4f_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4f_prof_sort -- python bench.py --phase sort --steps 3 --no-live-traffic > $O/r4f_prof_sort.log 2>&1
find $O/r4f_prof_sort -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r4f_sort_kernel_stats.csv
rm -rf $O/r4f_prof_sort
tail -n 4 $O/r4f_tests_new.txt $O/r4f_tests_parity.txt
cat $O/r4f_sort_ab.txt
