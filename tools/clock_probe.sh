#!/bin/bash
# Samples power / clocks while the bench kernel runs back to back.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rocm-smi --showmaxpower --showclocks 2>/dev/null | grep -v "^=\|^$" | head -20
python bench.py --steps 60000 --warmup 20 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err &
BP=$!
sleep 6
for i in 1 2 3 4; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|junction\|fclk" | head -8; echo --; sleep 1; done
wait $BP
cut -c1-160 /tmp/b.json
