#!/bin/bash
# phase stamps of ens_atoms_kernel in twin mode (mask coordinate on the targets) next to the plain launch, longitude-fastest
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for m in "" mask; do
  WBX_ALTERNATE_STREAMS=0 WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_eaprof.so WBX_EA_PROF_DUMP=/tmp/prof_$m.bin python tools/bench_ens_binned.py lon_fastest $m > /dev/null 2>&1
  echo "== lon_fastest ${m:-no mask}"; python tools/ea_prof.py /tmp/prof_$m.bin | head -24
done
