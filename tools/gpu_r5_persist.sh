#!/bin/bash
# (the persistent-wave flavour only exists in the diagnostic build: `make -C weatherbenchx_amd/csrc ab-eapersist` first;
#  WBX_ENS_ATOMS_PERSIST / WBX_ENS_ATOMS_STATIC are read by that build alone -- profiles/r05_ens_atoms_persistent_ab.txt)
export WBX_LIBRARY_PATH=${WBX_LIBRARY_PATH:-${GRAFT_REPO_ROOT:-$PWD}/weatherbenchx_amd/libwbx_hip_eapersist.so}
# Round 5: ens_atoms_kernel with persistent waves (per-XCD patch queues) against one block per patch (WBX_ENS_ATOMS_PERSIST=0),
# same box, alternating; parity first.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-a}
( timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_ens_binned.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5_pytest_persist_$TAG.log
tail -5 gpurun_out/r5_pytest_persist_$TAG.log
OUT=gpurun_out/r5_bench_persist_$TAG.txt
: > $OUT
for rep in 1 2; do
  for p in ${PERSIST_SET:-0 1}; do
    for what in "" mask nanmask skipna; do
      echo "== persist $p $what" >> $OUT
      WBX_ENS_ATOMS_PERSIST=$p timeout 300 python tools/bench_ens_binned.py lon_fastest $what 2>&1 | tail -2 >> $OUT
    done
  done
done
cat $OUT
