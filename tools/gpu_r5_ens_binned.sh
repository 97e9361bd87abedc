#!/bin/bash
# Round 5, first visit: parity of the widened wbx_ens_binned (per-point NaN masks, skipna) + the compact bench line on --small,
# then same-box A/B of ens_atoms_kernel against the round-4 library (weatherbenchx_amd/libwbx_hip_r4.so) and the new legs.
# (the A/B half needs weatherbenchx_amd/libwbx_hip_r4.so, a build of the round-4 commit, AND a binding without wbx_chunk_replay:
#  it ran before chunk records went in -- profiles/r05_ens_binned_ab.txt; today only the second half runs)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-a}
( timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_ens_binned.py tests/test_gpu_round4.py tests/test_bench_line.py tests/test_corners.py tests/test_foreign_arrays.py tests/test_loaders.py -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r5_pytest_ens_binned_$TAG.log
tail -15 gpurun_out/r5_pytest_ens_binned_$TAG.log
OUT=gpurun_out/r5_bench_ens_binned_$TAG.txt
: > $OUT
for rep in 1 2; do
for lay in lon_fastest lat_fastest; do
  for lib in libwbx_hip_r4.so libwbx_hip.so; do
    WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/$lib timeout 300 python tools/bench_ens_binned.py $lay >> $OUT 2>&1
    WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/$lib timeout 300 python tools/bench_ens_binned.py $lay mask >> $OUT 2>&1
  done
done
done
for lay in lon_fastest lat_fastest ifs; do
  timeout 300 python tools/bench_ens_binned.py $lay nanmask >> $OUT 2>&1
  timeout 300 python tools/bench_ens_binned.py $lay skipna >> $OUT 2>&1
  timeout 300 python tools/bench_ens_binned.py $lay nanmask skipna >> $OUT 2>&1
done
WBX_ENS_BINNED=0 timeout 300 python tools/bench_ens_binned.py lon_fastest nanmask >> $OUT 2>&1
WBX_ENS_BINNED=0 timeout 300 python tools/bench_ens_binned.py lon_fastest skipna >> $OUT 2>&1
cat $OUT
