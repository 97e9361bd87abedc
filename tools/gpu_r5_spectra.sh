#!/bin/bash
# Round 5: order-independent spectra sums -- reproducibility + parity, then the cost against the round-4 library (atomics).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-a}
( timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_spectra.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_golden.py -m gpu -x -q -k "spectr or spectra or det_spectrum or golden or config" 2>&1 | tail -30 ) > gpurun_out/r5_pytest_spectra_$TAG.log
tail -12 gpurun_out/r5_pytest_spectra_$TAG.log
OUT=gpurun_out/r5_bench_spectra_$TAG.txt
: > $OUT
for rep in 1 2; do
for lib in libwbx_hip.so; do  # (the round-4 library lacks wbx_chunk_replay: its numbers are profiles/r04_*)
  for lay in lon_fastest lat_fastest; do
    echo "== $lib $lay" >> $OUT
    WBX_CHUNK_REPLAY=0 WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/$lib timeout 300 python bench.py --legs spectrum --no-cpu --no-config5 --steps 10 --warmup 3 --layout $lay > /dev/null 2>>$OUT.err
    python - >> $OUT <<PY
import json
d=json.load(open('bench_full.json'))
s=d['spectrum']
print('spectrum ms_per_step', round(s['ms_per_step'],4), 'kernel_ms', s['roofline']['kernel_ms'], 'frac', s['roofline']['frac'],
      '| composite ms_per_chunk', round(s['with_deterministic_suite']['ms_per_chunk'],4), s['with_deterministic_suite'].get('roofline',{}).get('kernel_ms'))
PY
  done
done
done
cat $OUT
