#!/bin/bash
# Round 5: spectra records (ordered sums) against the round-4 tree (atomics) on the SAME box, alternating.
# ab_r4/ = `git archive <round-4 commit> | tar -x -C ab_r4` + its library as ab_r4/weatherbenchx_amd/libwbx_hip.so (made by hand in the build
# container for the measurement, not kept in the tree).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r5_spectra_ab.txt
: > $OUT
for rep in 1 2 3; do
for tree in ab_r4 .; do
  for lay in lon_fastest lat_fastest; do
    echo "== rep $rep tree $tree $lay" >> $OUT
    ( cd $R/$tree && rm -f bench_full.json && WBX_CHUNK_REPLAY=0 timeout 300 python bench.py --legs spectrum --no-cpu --no-config5 --steps 10 --warmup 3 --layout $lay > $R/gpurun_out/ab_line.json 2>>$OUT.err
      python - >> $OUT <<PY
import json, os
d = json.load(open('bench_full.json')) if os.path.exists('bench_full.json') else json.loads(open('$R/gpurun_out/ab_line.json').read().strip().splitlines()[-1])
s = d['spectrum']
w = s.get('with_deterministic_suite', {})
print('spectrum kernel_ms', s['roofline']['kernel_ms'], 'frac', s['roofline']['frac'], '| composite ms_per_chunk', round(w.get('ms_per_chunk', 0), 4), w.get('roofline', {}).get('kernel_ms'))
PY
    )
  done
done
done
cat $OUT
