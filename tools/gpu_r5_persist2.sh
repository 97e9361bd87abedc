#!/bin/bash
# (the persistent-wave flavour only exists in the diagnostic build: `make -C weatherbenchx_amd/csrc ab-eapersist` first;
#  WBX_ENS_ATOMS_PERSIST / WBX_ENS_ATOMS_STATIC are read by that build alone -- profiles/r05_ens_atoms_persistent_ab.txt)
export WBX_LIBRARY_PATH=${WBX_LIBRARY_PATH:-${GRAFT_REPO_ROOT:-$PWD}/weatherbenchx_amd/libwbx_hip_eapersist.so}
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
