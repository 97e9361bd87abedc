#!/bin/bash
# Same-box A/B of the mean shift in front of the fp32 1440-point transforms (WBX_SPECTRUM_DEMEAN, csrc/wbx_zspec1440.hpp):
# libwbx_hip.so against `make ab-nodemean`'s libwbx_hip_nodemean.so -- accuracy against the float64 oracle
# (tests/measure_spectrum_error.py) and the spectrum legs of bench.py on both layouts.  Output: gpurun_out/demean_ab.txt
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/demean_ab.txt
: > $OUT
for lib in libwbx_hip_nodemean.so libwbx_hip.so; do
  export WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/$lib
  echo "=== $lib" >> $OUT
  timeout 120 python tests/measure_spectrum_error.py >> $OUT 2>&1
done
for rep in 1 2; do
  for lib in libwbx_hip_nodemean.so libwbx_hip.so; do
    export WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/$lib
    for layout in lon_fastest lat_fastest; do
      timeout 200 python bench.py --legs spectrum --layout $layout --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['spectrum']; w=d['with_deterministic_suite']
print('$lib $layout rep$rep: spectrum kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], '| composite ms_per_chunk', round(w['ms_per_chunk'],4), 'kernel_ms', w.get('roofline',{}).get('kernel_ms'), 'sumS', d['check']['sum_k_S_k'])" >> $OUT 2>&1
    done
  done
done
cat $OUT
