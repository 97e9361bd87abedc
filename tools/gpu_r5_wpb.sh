#!/bin/bash
# Round 5: ens_atoms_kernel on latitude-fastest public chunks (721-point rows): a block barrier every 2 / 4 / 16 rows among the four waves of a block (make ab-eabar2 ...; waves per block 6 / 12 were measured first: slower, same FETCH_SIZE)
# (make ab-eawpb6 ab-eawpb12), same box, alternating; FETCH_SIZE of each in a separate pass.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_wpb.txt
: > $OUT
for rep in 1 2; do
for v in default eabar1 eabar2; do
  if [ $v = default ]; then unset WBX_LIBRARY_PATH; else export WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/libwbx_hip_$v.so; fi
  for what in "" nanmask; do
    echo "== $v $what" >> $OUT
    timeout 300 python tools/bench_ens_binned.py lat_fastest $what 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_chunk', d['ms_per_chunk'], 'kernel_ms', d['kernel_ms_per_chunk'], 'crps', d['crps_global'])" >> $OUT
  done
done
done
unset WBX_LIBRARY_PATH
cd /tmp
for v in default eabar1 eabar2; do
  if [ $v = default ]; then unset WBX_LIBRARY_PATH; else export WBX_LIBRARY_PATH=$GRAFT_REPO_ROOT/weatherbenchx_amd/libwbx_hip_$v.so; fi
  rm -rf /tmp/pmc_$v
  timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_$v -o r1 -- python $GRAFT_REPO_ROOT/tools/bench_ens_binned.py lat_fastest > /dev/null 2>&1
  python - >> $OUT <<PY
import sqlite3, glob
for db in glob.glob('/tmp/pmc_$v/*/r1_results.db') + glob.glob('/tmp/pmc_$v/r1_results.db'):
  for k, c, n, v in sqlite3.connect(db).execute("select substr(kernel_name, 1, 40), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%ens_atoms_kernel%' group by 1, 2"):
    print('$v', k, c, n, 'FETCH x2 bytes', v * 1024 * 2, 'ratio to 1.7276 GB', round(v * 1024 * 2 / 1727631360, 4))
PY
done
cat $OUT
