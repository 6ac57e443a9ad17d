#!/bin/bash
# round 5, session p: the full GPU suite after the staging thresholds moved; the driver-form bench line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5p}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|ERROR|Error" | tail -30 > $O/tests.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err
cat $O/tests.txt; tail -1 $O/build.txt; tail -2 $O/bench.err
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_driver.json") if l.startswith("{")][-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("frac", "kernel_ms", "frac_same_buffers", "no_bounds_ms", "frac_moved")})
print({k: (v.get("ms_per_call"), v.get("bit_exact")) for k, v in d["configs"].items() if isinstance(v, dict)})
print([(x["partitions"], x["gpu_call_us"], x["cpu_oracle_us"]) for x in d["small_call"]["rows"]], [(x["partitions"], x["grouped_us"], x["cpu_oracle_us"]) for x in (d["small_call"].get("c_abi") or {}).get("rows", [])])
hb = d["host_boundary"]; print({k: hb.get(k) for k in ("ms", "pinned_ms", "sparse_begin_ms", "grouped_ms")})
PY
