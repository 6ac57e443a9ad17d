#!/bin/bash
# ONE parameterised GPU-box session (replaces the per-session gpu_session_r4*/r5* scripts of rounds 4-5).
#
#   gpurun --timeout 1500 -- 'tools/gpu_session.sh TAG STEP [STEP ...]'
#
# Everything lands under gpurun_out/TAG/ (scratch); tools/publish_profiles.py copies what should be judged into profiles/.
# Steps (run in the order given):
#   build      __graft_entry__.build() + smoke()
#   tests      pytest -m gpu (TESTS_K='-k expr' narrows it; TESTS_FILES='tests/a.py tests/b.py')
#   stress     tools/stress_gpu.py (STRESS_ARGS, STRESS_SEED0)
#   bench      bench.py --gpus 1 --steps 20 --warmup 5  (the driver's command)  -> bench_driver.json + a digest
#   benchlong  bench.py (default 1000 steps)
#   stats      rocprofv3 --kernel-trace --stats of the driver's bench command (no oracle legs)
#   pmc        FETCH_SIZE / WRITE_SIZE / SQ / LDS counter passes over tools/pmc_probe.py (PROBE_ARGS) -> pmc_summary.json
#   cfg5       tools/cfg5_probe.py + rocprofv3 stats of the cfg5 call (large_probe) ; CFG5_ARGS for other sizes
#   block      tools/block_probe.py + rocprofv3 stats of cfg2b and two block batches
#   blockpmc   LDS / VALU counter passes over the cfg2b call
#   sort       bench.py --phase sort + rocprofv3 stats / timeline of the 33.5 M-partition sort
#   latency    tools/latency_probe.py + tools/latency_c.c at the C ABI
#   lab:CMD    any command (quoted), output to lab_N.txt  -- for one-off A/B runs, e.g. "lab:LA_X=1 python tools/cfg5_probe.py"
set -u
TAG=${1:-r06}; shift || true
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
LAB=0
prof() { ( cd /tmp && timeout ${PROF_TIMEOUT:-600} rocprofv3 "$@" ); }
trim() { find $O -name "*.db" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null; }
for STEP in "$@"; do
  echo "== $STEP"
  case "$STEP" in
    build)
      python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt; tail -2 $O/build.txt ;;
    tests)
      timeout ${TESTS_TIMEOUT:-1800} python -m pytest ${TESTS_FILES:-tests} -q -m gpu ${TESTS_K:-} -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
      grep -E "passed|failed|FAILED|rror" $O/pytest.log | tail -12 ;;
    stress)
      STRESS_SEED0=${STRESS_SEED0:-61000} timeout 900 python tools/stress_gpu.py ${STRESS_ARGS:-100 40 60 40 10 10 300} 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/stress.txt ;;
    bench)
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err; tail -2 $O/bench.err
      python tools/bench_digest.py $O/bench_driver.json | tee $O/bench_digest.txt ;;
    benchlong)
      timeout 600 python bench.py > $O/bench.json 2>> $O/bench.err; python tools/bench_digest.py $O/bench.json ;;
    stats)
      prof --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sort-phase > $O/stats.log 2>&1
      find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-200 ;;
    pmc)
      PROBE="python $R/tools/pmc_probe.py ${PROBE_ARGS:-}"
      prof --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PROBE > $O/pmc_fetch.log 2>&1
      prof --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PROBE > $O/pmc_write.log 2>&1
      prof --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $PROBE > $O/pmc_sq.log 2>&1
      prof --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- $PROBE > $O/pmc_sq2.log 2>&1
      python tools/pmc_parse.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 > $O/pmc_summary.json 2> $O/pmc_parse.err
      head -c 2500 $O/pmc_summary.json; echo ;;
    cfg5)
      timeout 600 python tools/cfg5_probe.py ${CFG5_ARGS:-} 2>&1 | grep -v amdgpu.ids | tee $O/cfg5_probe.txt
      prof --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -- python $R/tools/large_probe.py --partitions ${CFG5_P:-1048576} --consumers ${CFG5_C:-8192} --launches 8 --dist pareto > $O/stats_cfg5.log 2>&1
      find $O/stats_cfg5 -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200 ;;
    block)
      LAG_BITS=40 timeout 300 python tools/block_probe.py 1,10000,128 200,8000,16 1000,2000,100 2>&1 | grep -v amdgpu.ids | tee $O/block_probe.txt
      LAG_BITS=40 prof --kernel-trace --stats --output-format csv -d $O/stats_block -- python $R/tools/block_probe.py 1,10000,128 200,8000,16 1000,2000,100 > $O/stats_block.log 2>&1
      find $O/stats_block -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-200 ;;
    blockpmc)
      LAG_BITS=40 prof --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_block -- python $R/tools/block_probe.py 1,10000,128 > $O/pmc_block.log 2>&1
      python tools/pmc_parse.py $O/pmc_block > $O/pmc_block_summary.json 2> $O/pmc_block_parse.err; head -c 1500 $O/pmc_block_summary.json; echo ;;
    sort)
      timeout 300 python bench.py --phase sort > $O/bench_sort.json 2>> $O/bench.err; cut -c1-900 $O/bench_sort.json
      prof --kernel-trace --stats --output-format csv -d $O/stats_sort -- python $R/tools/large_probe.py --partitions 33554432 --consumers 0 --launches 8 > $O/stats_sort.log 2>&1
      python3 tools/trace_timeline.py $O/stats_sort > $O/sort_timeline.txt 2>&1; tail -30 $O/sort_timeline.txt ;;
    latency)
      timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/latency.txt
      gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$R/kafka_lag_based_assignor_amd -o /tmp/latency_c && timeout 120 /tmp/latency_c oracle/liblagoracle.so 2>&1 | tee $O/latency_c.txt ;;
    lab:*)
      LAB=$((LAB + 1)); echo "${STEP#lab:}" > $O/lab_$LAB.txt
      ( eval "timeout ${LAB_TIMEOUT:-600} ${STEP#lab:}" ) 2>&1 | grep -v amdgpu.ids | tee -a $O/lab_$LAB.txt | tail -${LAB_TAIL:-40} ;;
    *) echo "unknown step $STEP" ;;
  esac
done
trim
du -sh $O
