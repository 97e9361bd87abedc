#!/bin/bash
# (the persistent-wave flavour only exists in the diagnostic build: `make -C weatherbenchx_amd/csrc ab-eapersist` first;
#  WBX_ENS_ATOMS_PERSIST / WBX_ENS_ATOMS_STATIC are read by that build alone -- profiles/r05_ens_atoms_persistent_ab.txt)
export WBX_LIBRARY_PATH=${WBX_LIBRARY_PATH:-${GRAFT_REPO_ROOT:-$PWD}/weatherbenchx_amd/libwbx_hip_eapersist.so}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for p in 0 1; do
  WBX_ENS_ATOMS_PERSIST=$p WBX_ALTERNATE_STREAMS=0 WBX_ALTERNATE_CHUNKS=0 WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_eaprof.so WBX_EA_PROF_DUMP=/tmp/prof_$p.bin python tools/bench_ens_binned.py lon_fastest > /dev/null 2>&1
  echo "== persist $p"; python tools/ea_prof.py /tmp/prof_$p.bin
done
