#!/bin/bash
# Round 4: wbx_ens_binned parity + timing after the register diet (no scratch, scalar row lookups).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-b}
( timeout 900 python -m pytest tests/test_ens_binned.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r4_pytest_ens_binned_$TAG.log
tail -5 gpurun_out/r4_pytest_ens_binned_$TAG.log
OUT=gpurun_out/r4_bench_ens_binned_$TAG.txt
: > $OUT
for lay in lon_fastest lat_fastest ifs; do
  timeout 300 python tools/bench_ens_binned.py $lay 2>&1 | grep -v amdgpu.ids >> $OUT
done
timeout 300 python tools/bench_ens_binned.py lon_fastest mask 2>&1 | grep -v amdgpu.ids >> $OUT
for rows in 8 24 32 48 64; do
  for lay in lon_fastest lat_fastest; do
    WBX_ENS_ATOMS_ROWS=$rows timeout 300 python tools/bench_ens_binned.py $lay 2>&1 | grep -v amdgpu.ids >> $OUT
  done
done
for lay in lon_fastest lat_fastest; do
  WBX_ENS_ATOMS_NT=1 timeout 300 python tools/bench_ens_binned.py $lay 2>&1 | grep -v amdgpu.ids >> $OUT
  WBX_ENS_ATOMS_NT=0 timeout 300 python tools/bench_ens_binned.py $lay 2>&1 | grep -v amdgpu.ids >> $OUT
  WBX_PATCH_ORDER=0 timeout 300 python tools/bench_ens_binned.py $lay 2>&1 | grep -v amdgpu.ids >> $OUT
  WBX_PATCH_ORDER=1 timeout 300 python tools/bench_ens_binned.py $lay 2>&1 | grep -v amdgpu.ids >> $OUT
done
python - <<PY
import json
for line in open('gpurun_out/r4_bench_ens_binned_TAG.txt'.replace('TAG', '$TAG')) if False else open('$OUT'):
  try: d = json.loads(line)
  except Exception: print(line.rstrip()); continue
  print(d['layout'], 'mask' if d['mask'] else '', 'rows', d['rows'], 'ms/launch', d['ms_per_launch'], 'frac', d['frac_of_hbm_peak_per_launch'], 'ms/chunk', d['ms_per_chunk'])
PY
