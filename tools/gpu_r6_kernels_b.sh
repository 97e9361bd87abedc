#!/bin/bash
# Round 6, same box: (1) det_atoms_kernel against its two timing diagnostics -- no hit / miss bookkeeping (diag-atomsknock), no
# flushes inside a sweep (diag-atomsnoflush) -- WRONG sums, times only; (2) counters of the fused det + spectra sweep;
# (3) ens_atoms_kernel on latitude-fastest chunks with a block barrier every 4 rows (make ab-eabar4): time and FETCH_SIZE.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
O=$R/gpurun_out/r6_atoms_diag.jsonl; : > $O
for v in "" atomsknock atomsnoflush "" atomsknock atomsnoflush; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_binned_ab.py "${v:-r6}" 2>/dev/null | tee -a $O
done
bash tools/pmc_det_spectrum.sh > $R/gpurun_out/r6_pmc_det_spectrum.txt 2>&1
tail -12 $R/gpurun_out/r6_pmc_det_spectrum.txt
cd $R
E=$R/gpurun_out/r6_ens_rowbarrier.txt; : > $E
for rep in 1 2; do
for v in default eabar4; do
  if [ $v = default ]; then unset WBX_LIBRARY_PATH; else export WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_$v.so; fi
  for what in "" mask nanmask; do
    echo "== $v $what" >> $E
    timeout 300 python tools/bench_ens_binned.py lat_fastest $what 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_chunk', d['ms_per_chunk'], 'kernel_ms', d['kernel_ms_per_chunk'], 'crps', d['crps_global'])" >> $E
  done
done
done
unset WBX_LIBRARY_PATH
cd /tmp
for v in default eabar4; do
  if [ $v = default ]; then unset WBX_LIBRARY_PATH; else export WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_$v.so; fi
  for what in "" nanmask; do
  rm -rf /tmp/pmc_$v
  timeout 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_$v -o r1 -- python $R/tools/bench_ens_binned.py lat_fastest $what > /dev/null 2>&1
  python - >> $E <<PY
import sqlite3, glob
for db in glob.glob('/tmp/pmc_$v/*/r1_results.db') + glob.glob('/tmp/pmc_$v/r1_results.db'):
  for k, c, n, v in sqlite3.connect(db).execute("select substr(kernel_name, 1, 40), counter_name, count(*), avg(value) from counters_collection where kernel_name like '%ens_atoms_kernel%' group by 1, 2"):
    print('$v', '$what', k, c, n, 'FETCH x2 bytes', v * 1024 * 2, 'ratio to 1.7276 GB', round(v * 1024 * 2 / 1727631360, 4))
PY
  done
done
cat $E
