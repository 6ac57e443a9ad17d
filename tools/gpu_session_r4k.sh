#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build()" > $O/r4k_build.txt 2>&1
for rep in 1 2; do
  python tools/group_probe.py >> $O/r4k_group.txt 2>&1
  LA_LIB_PATH=$PWD/tools/_lab/group_sub1.so python tools/group_probe.py >> $O/r4k_group.txt 2>&1
  LA_LIB_PATH=$PWD/tools/_lab/group_sub4.so python tools/group_probe.py >> $O/r4k_group.txt 2>&1
  LA_NO_SMALL_GROUP=1 python tools/group_probe.py >> $O/r4k_group.txt 2>&1
done
grep -v amdgpu.ids $O/r4k_group.txt
