#!/bin/bash
# Round 5: WBX_BINNED_ACCUMULATE (the binned launches add into the chunk loop's accumulator themselves) against scratch + wbx_acc_add,
# same box, alternating: chunk-loop ms per chunk of the public-benchmark chunks.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r5_fused_add.txt
: > $OUT
for rep in 1 2 3; do
for f in 0 1; do
  for lay in lon_fastest lat_fastest; do
    WBX_FUSED_ACC_ADD=$f timeout 300 python bench.py --legs public_chunk,public_chunk_ens --no-cpu --no-config5 --steps 5 --warmup 2 --layout $lay > /dev/null 2>>$OUT.err
    python - >> $OUT <<PY
import json
d = json.load(open('bench_full.json'))
pc, p = d['public_chunk'], d['public_chunk_ens']
print('fused $f $lay pc', round(pc['ms_per_chunk'], 4), pc.get('chunk_over_kernel'), '| pce', round(p['ms_per_chunk'], 4), p.get('chunk_over_kernel'),
      '| mask', round(p['with_mask_coordinate']['ms_per_chunk'], 4), '| nanmask', round(p['with_nan_mask']['ms_per_chunk'], 4))
PY
  done
done
done
cat $OUT
