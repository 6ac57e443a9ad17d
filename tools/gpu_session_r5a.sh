#!/bin/bash
# round 5, session a: hints through the host entry points, the sentinel fix, the rotating bench -- build + smoke, the new tests first, the full GPU suite, the driver's bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r5a}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_new.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -8 > $O/tests.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cat $O/tests_new.txt $O/tests.txt; tail -1 $O/build.txt; tail -3 $O/bench.err; python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"])
    print({k: v.get("ms_per_call") for k, v in d["configs"].items() if isinstance(v, dict)})
    print(d["small_call"]["rows"])
    hb = d["host_boundary"]; print({k: hb.get(k) for k in ("ms", "pinned_ms", "hint", "sparse_begin_ms", "grouped_ms")})
    print(d["sort_phase"]["kernel_ms"], d["sort_phase"]["frac"])
except Exception as e:
    print("bench parse:", e)
PY
