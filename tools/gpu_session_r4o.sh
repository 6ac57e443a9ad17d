#!/bin/bash
# round 4, session o: the moved-bins sort of the greedy rounds (cfg5's chain)
mkdir -p gpurun_out/r4o
python tools/cfg5_probe.py --reps 10 > gpurun_out/r4o/cfg5_probe.txt 2>&1
LA_LIB_PATH=tools/_lab/clocks.so python tools/cfg5_probe.py --reps 4 > gpurun_out/r4o/cfg5_clocks.txt 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -k "sort_only_the_bins_that_move or merge_ascending_runs or cfg5 or large" 2>&1 | tail -5 > gpurun_out/r4o/tests.txt
tail -3 gpurun_out/r4o/tests.txt; cat gpurun_out/r4o/cfg5_probe.txt gpurun_out/r4o/cfg5_clocks.txt
