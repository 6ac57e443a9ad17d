#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
python -c "import __graft_entry__ as g; g.build()" > $O/r4g_build.txt 2>&1
timeout 600 python tools/mapped_probe.py > $O/r4g_mapped.txt 2>&1
for cp in 700000 2800000 5600000; do
  echo "LA_CHUNK_PARTITIONS=$cp" >> $O/r4g_mapped.txt
  LA_CHUNK_PARTITIONS=$cp timeout 600 python tools/mapped_probe.py 2>&1 | grep "la_assign_batch" >> $O/r4g_mapped.txt
done
cat $O/r4g_mapped.txt
