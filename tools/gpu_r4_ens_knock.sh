#!/bin/bash
# Same-box A/B: wbx_ens_binned as shipped, its knock-out builds (make ab-eak1 ab-eak2) and the un-binned pipelined kernel
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lay in lon_fastest lat_fastest; do
  bash tools/trace_ens_binned.sh $lay 2>&1 | grep "ens_atoms\|ens_pipe"
  WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_eak1.so bash tools/trace_ens_binned.sh $lay 2>&1 | grep "ens_atoms\|ens_pipe" | sed 's/^/eak1 /'
  WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_eak2.so bash tools/trace_ens_binned.sh $lay 2>&1 | grep "ens_atoms\|ens_pipe" | sed 's/^/eak2 /'
  WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_eak2.so WBX_ENS_ATOMS_ROWS=48 bash tools/trace_ens_binned.sh $lay 2>&1 | grep "ens_atoms\|ens_pipe" | sed 's/^/eak2 rows48 /'
  WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_eak2.so WBX_ENS_ATOMS_ROWS=8 bash tools/trace_ens_binned.sh $lay 2>&1 | grep "ens_atoms\|ens_pipe" | sed 's/^/eak2 rows8 /'
done
cd /tmp
for lay in lon_fastest lat_fastest; do
OUT=$R/gpurun_out/trace_nobins_$lay; rm -rf $OUT; mkdir -p $OUT
WBX_ALTERNATE_STREAMS=0 timeout 200 rocprofv3 --kernel-trace -d $OUT/a -o t --output-format csv -- python $R/tools/bench_ens_binned.py $lay nobins > $OUT/a.log 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$OUT/a/*kernel_trace.csv')):
  dur = collections.defaultdict(list)
  for row in csv.DictReader(open(f)):
    dur[row['Kernel_Name'][:60]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
  for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:3]:
    v = sorted(v)
    print('nobins $lay', k, 'n', len(v), 'avg_us', round(sum(v) / len(v), 1), 'median_us', round(v[len(v) // 2], 1))
PY
done
