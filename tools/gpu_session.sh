#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 stats + PMC passes.  Usage: tools/gpu_session.sh TAG
# Everything lands under gpurun_out/TAG/ (scratch; copy what should be judged into profiles/).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
  tail -5 $O/pytest.log
fi
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.json
# what the driver runs (20 timed steps after 5 warm-up steps): must agree with the long run within a few percent
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2>> $O/bench.err; tail -1 $O/bench_driver.json | cut -c1-600
timeout 300 python bench.py --phase sort > $O/bench_sort.json 2>> $O/bench.err; cat $O/bench_sort.json | cut -c1-900
# the N>1 code path (RCCL init, la_plan_shards split, all-gather inside the timed region, max-reduce) at world size 1
LA_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --scaling strong --workload cfg4 --steps 200 --warmup 20 --no-sort-phase > $O/bench_strong_cfg4.json 2>> $O/bench.err; tail -1 $O/bench_strong_cfg4.json | cut -c1-700
timeout 300 python bench.py --reset-mode latest --no-cpu-baseline --no-sort-phase > $O/bench_latest.json 2>> $O/bench.err
timeout 300 python bench.py --algo wide --no-cpu-baseline --no-sort-phase > $O/bench_wide.json 2>> $O/bench.err
timeout 300 python bench.py --workload cfg4 --no-cpu-baseline --no-sort-phase > $O/bench_cfg4.json 2>> $O/bench.err
cat $O/bench_latest.json $O/bench_wide.json $O/bench_cfg4.json | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-sort-phase > $O/stats.log 2>&1
PROBE="python $R/tools/pmc_probe.py $PROBE_ARGS"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PROBE > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PROBE > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $PROBE > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- $PROBE > $O/pmc_sq2.log 2>&1
cd $R
python tools/pmc_parse.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 > $O/pmc_summary.json 2> $O/pmc_parse.err
# keep the merge small: drop bulky per-dispatch traces, keep stats + counter CSVs
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh $O
head -c 3000 $O/pmc_summary.json
