#!/bin/bash
# round 5, session ag: the round's last build -- build + smoke, the full GPU suite, stress, the driver's bench, kernel stats of the bench command
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5ag}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|error|Error" | tail -8 > $O/tests.txt
STRESS_SEED0=51000 timeout 600 python tools/stress_gpu.py 100 40 60 40 10 10 300 2>&1 | grep -v amdgpu.ids | tail -3 > $O/stress.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sort-phase > $O/stats.log 2>&1
LAG_BITS=40 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_block -- python $R/tools/block_probe.py 1,10000,128 200,8000,16 1000,2000,100 > $O/stats_block.log 2>&1
cd $R
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/latency.txt
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -ldl -Wl,-rpath,$R/kafka_lag_based_assignor_amd -o /tmp/latency_c && timeout 120 /tmp/latency_c oracle/liblagoracle.so > $O/latency_c.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
cat $O/tests.txt $O/stress.txt; tail -1 $O/build.txt; tail -2 $O/bench.err
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_driver.json") if l.startswith("{")][-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("frac", "kernel_ms", "frac_same_buffers", "no_bounds_ms", "frac_moved")}, r.get("wire_out", {}).get("fused_ms"))
print({k: (v.get("ms_per_call"), v.get("bit_exact"), v.get("sha256_matches_frozen_literal_oracle")) for k, v in d["configs"].items() if isinstance(v, dict)})
print([(x["partitions"], x["gpu_call_us"], x["cpu_oracle_us"]) for x in d["small_call"]["rows"]], [(x["partitions"], x["grouped_us"], x["cpu_oracle_us"]) for x in (d["small_call"].get("c_abi") or {}).get("rows", [])])
print(d["sort_phase"].get("kernel_ms"), d["sort_phase"].get("frac"), d["parity"], d["cpu_baseline"]["value"])
hb = d["host_boundary"]; print({k: hb.get(k) for k in ("ms", "pinned_ms", "sparse_begin_ms", "grouped_ms")})
PY
