#!/bin/bash
# Round 4, first visit: parity of wbx_ens_binned (small + full size), then its timing on the public probabilistic chunk.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-a}
( timeout 900 python -m pytest tests/test_ens_binned.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r4_pytest_ens_binned_$TAG.log
tail -15 gpurun_out/r4_pytest_ens_binned_$TAG.log
OUT=gpurun_out/r4_bench_ens_binned_$TAG.txt
: > $OUT
for lay in lon_fastest lat_fastest ifs; do
  timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
  timeout 300 python tools/bench_ens_binned.py $lay mask >> $OUT 2>&1
done
for rows in 8 12 24 32 48; do
  for lay in lon_fastest lat_fastest; do
    WBX_ENS_ATOMS_ROWS=$rows timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
  done
done
for lay in lon_fastest lat_fastest; do
  WBX_ENS_BINNED=0 timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
  WBX_ENS_ATOMS_NT=1 timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
  WBX_ENS_ATOMS_NT=0 timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
  WBX_PATCH_ORDER=0 timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
  WBX_PATCH_ORDER=1 timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
done
cat $OUT
