#!/bin/bash
# tools/ubench/flat_fetch.hip under the L2 request counters: buffer aligned / shifted by 64 bytes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/flatfetch; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/flat_fetch.hip -o /tmp/flat_fetch 2>/dev/null
for shift in 0 16 32; do
  /tmp/flat_fetch $shift
  timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/s$shift -o r -- /tmp/flat_fetch $shift > /dev/null 2>&1
  python - <<PY
import sqlite3
rows = sqlite3.connect('$O/s$shift/r_results.db').execute("select substr(kernel_name, 1, 40), counter_name, count(*), avg(value) from counters_collection group by 1, 2").fetchall()
for k, c, n, v in rows: print('   shift $shift', k, c, n, '%.0f' % v)
PY
done
rm -rf $O
