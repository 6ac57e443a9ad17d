#!/bin/bash
# round 5, session ai: the driver-form bench line after the block kernel's two-stage lag
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
O=$R/gpurun_out/${1:-r5ai}
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build.txt 2>&1; echo "smoke rc=$?" >> $O/build.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-sort-phase > $O/bench_driver.json 2> $O/bench.err
tail -1 $O/build.txt
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_driver.json") if l.startswith("{")][-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], {k: r.get(k) for k in ("frac", "kernel_ms")})
print({k: (v.get("ms_per_call"), v.get("bit_exact"), v.get("sha256_matches_frozen_literal_oracle")) for k, v in d["configs"].items() if isinstance(v, dict)})
PY
