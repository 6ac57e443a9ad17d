#!/bin/bash
# round 4, session u: kernel trace and counters of the block path, network against digits (LA_BLOCK_RADIX=0 / 2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4u; mkdir -p $O
SH="200,8000,16 1000,2000,100 1,10000,128"
for m in 0 2; do
  LA_BLOCK_RADIX=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$m -- python tools/block_probe.py $SH > $O/stats$m.log 2>&1
  f=$(find $O/stats$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/block_kernel_stats_radix$m.csv
  LA_BLOCK_RADIX=$m timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc$m -- python tools/block_probe.py 200,8000,16 > $O/pmc$m.log 2>&1
  python tools/pmc_parse.py $O/pmc$m > $O/block_pmc_radix$m.json 2>&1
  rm -rf $O/stats$m $O/pmc$m
done
head -4 $O/block_kernel_stats_radix0.csv $O/block_kernel_stats_radix2.csv | cut -c1-160; head -c 1500 $O/block_pmc_radix0.json; echo; head -c 1500 $O/block_pmc_radix2.json
