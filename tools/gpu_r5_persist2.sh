#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r5_bench_persist_grid.txt
: > $OUT
for p in 0 1 2048 3072 6144 16000 0 1; do
  echo "== persist $p" >> $OUT
  WBX_ENS_ATOMS_PERSIST=$p timeout 300 python tools/bench_ens_binned.py lon_fastest 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_chunk'], d['kernel_ms_per_chunk'])" >> $OUT
done
cat $OUT
