export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu -k "key32 or 32_bit or forms_agree" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "block" 2>&1 | tail -2
SH="1,10000,128 1,8000,256 1,16000,200 1,3000,100 600,300,128 2000,1000,200"
for m in 1 0 1; do echo "== LA_BLOCK_KEY32=$m"; LA_BLOCK_KEY32=$m timeout 300 python tools/block_probe.py $SH 2>&1 | grep -v amdgpu | cut -c1-100; done
echo "== clocks"; LA_LIB_PATH=$PWD/tools/_lab/block_clocks.so timeout 200 python tools/block_probe.py 1,10000,128 1,8000,256 1,16000,200 2>&1 | grep -v amdgpu
