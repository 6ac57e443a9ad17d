#!/bin/bash
# round 5, session j: the driver's bench + rocprofv3 kernel stats of the same command + PMC passes + block-path stats -> profiles/r05_*
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5j}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err
timeout 400 python bench.py > $O/bench.json 2>> $O/bench.err
LA_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --scaling strong --workload cfg4 --steps 200 --warmup 20 --no-sort-phase > $O/bench_strong_cfg4.json 2>> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sort-phase > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_block -- python $R/tools/block_probe.py 1,10000,128 200,8000,16 1000,2000,100 > $O/stats_block.log 2>&1
PROBE="python $R/tools/pmc_probe.py"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PROBE > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PROBE > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- $PROBE > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- $PROBE > $O/pmc_sq2.log 2>&1
cd $R
python tools/pmc_parse.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 > $O/pmc_summary.json 2> $O/pmc_parse.err
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh $O; tail -1 $O/build.txt; tail -2 $O/bench.err
python - <<PY
import json
for f in ("bench_driver", "bench"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f) if l.startswith("{")][-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], {k: r.get(k) for k in ("frac", "kernel_ms", "frac_same_buffers", "no_bounds_ms", "frac_moved", "wire_out")})
        if d.get("configs"): print({k: (v.get("ms_per_call"), v.get("bit_exact")) for k, v in d["configs"].items() if isinstance(v, dict)})
        if d.get("small_call"): print(d["small_call"]["rows"][:3], (d["small_call"].get("c_abi") or {}).get("rows"))
        if d.get("sort_phase"): print(d["sort_phase"].get("kernel_ms"), d["sort_phase"].get("frac"))
    except Exception as e:
        print(f, "parse:", e)
PY
