#!/bin/bash
# round 5, session d: the generated 32-bit sort network in the block path's greedy -- tests, A/B against LA_BLOCK_KEY32=0
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r5d}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu -k "key32 or 32_bit or forms_agree" 2>&1 | tail -15 > $O/tests_new.txt
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "block" 2>&1 | tail -5 >> $O/tests_new.txt
SH="1,10000,128 1,8000,256 1,16000,200 1,3000,100 600,300,128 2000,1000,200"
for m in 1 0 1 0; do
  echo "== LA_BLOCK_KEY32=$m" >> $O/block_ab.txt
  LA_BLOCK_KEY32=$m timeout 300 python tools/block_probe.py $SH 2>&1 | grep -v amdgpu | cut -c1-100 >> $O/block_ab.txt
done
cat $O/tests_new.txt; tail -1 $O/build.txt; cat $O/block_ab.txt
