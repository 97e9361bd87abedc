#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lay in lon_fastest lat_fastest; do
  WBX_ALTERNATE_STREAMS=0 WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip_eaprof.so WBX_EA_PROF_DUMP=/tmp/prof_$lay.bin python tools/bench_ens_binned.py $lay > /dev/null 2>&1
  echo "== $lay"; python tools/ea_prof.py /tmp/prof_$lay.bin
done
