#!/bin/bash
# (the persistent-wave flavour only exists in the diagnostic build: `make -C weatherbenchx_amd/csrc ab-eapersist` first;
#  WBX_ENS_ATOMS_PERSIST / WBX_ENS_ATOMS_STATIC are read by that build alone -- profiles/r05_ens_atoms_persistent_ab.txt)
export WBX_LIBRARY_PATH=${WBX_LIBRARY_PATH:-${GRAFT_REPO_ROOT:-$PWD}/weatherbenchx_amd/libwbx_hip_eapersist.so}
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r5_bench_persist_variants.txt
: > $OUT
run() { echo "== $1" >> $OUT; env $2 timeout 300 python tools/bench_ens_binned.py lon_fastest 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_chunk'], d['kernel_ms_per_chunk'])" >> $OUT; }
for rep in 1 2; do
run "plain" "WBX_ENS_ATOMS_PERSIST=0"
run "persist queue" "WBX_ENS_ATOMS_PERSIST=1"
run "persist static" "WBX_ENS_ATOMS_PERSIST=1 WBX_ENS_ATOMS_STATIC=1"
run "noreload queue" "WBX_ENS_ATOMS_PERSIST=1 WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/libwbx_hip_eanoreload.so"
run "noreload static" "WBX_ENS_ATOMS_PERSIST=1 WBX_ENS_ATOMS_STATIC=1 WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/libwbx_hip_eanoreload.so"
done
cat $OUT
