#!/bin/bash
# round 4, session p: full GPU suite after the moved-bins sort; cfg5 with the keys-first order of passes; stress
mkdir -p gpurun_out/r4p
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4p/tests.txt
for kf in 0 1; do echo "LA_SORT_KEYS_FIRST=$kf"; LA_SORT_KEYS_FIRST=$kf python tools/cfg5_probe.py --reps 10 2>&1 | grep -E "^default"; done > gpurun_out/r4p/cfg5_keys_first.txt
python tools/stress_gpu.py 10 60 10 120 10 10 > gpurun_out/r4p/stress.txt 2>&1
tail -3 gpurun_out/r4p/tests.txt; cat gpurun_out/r4p/cfg5_keys_first.txt; tail -5 gpurun_out/r4p/stress.txt
