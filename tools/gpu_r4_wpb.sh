#!/bin/bash
# round 4: ens_atoms_kernel with 4 waves per block on adjacent x tiles of 721-point rows against lone waves (make ab-eawpb1),
# same box: time of the launch on the latitude-fastest public chunk (unmasked / masked twin), FETCH_SIZE
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
for rep in 1 2 3; do
  bash tools/trace_ens_binned.sh lat_fastest 2>&1 | grep "ens_atoms" | sed "s/^/wpb4 /"
  WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/libwbx_hip_eawpb1.so bash tools/trace_ens_binned.sh lat_fastest 2>&1 | grep "ens_atoms" | sed "s/^/wpb1 /"
done
for lib in "" _eawpb1; do
  [ -n "$lib" ] && export WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/libwbx_hip$lib.so
  timeout 300 python bench.py --legs public_chunk_ens --no-cpu --no-config5 --steps 20 --warmup 5 --layout lat_fastest 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().split('\n')[-1])['public_chunk_ens']
print('lib$lib: unmasked ms/chunk', round(r['ms_per_chunk'], 4), 'kernel', r['roofline']['kernel_ms'], '| masked ms/chunk', round(r['with_mask_coordinate']['ms_per_chunk'], 4), 'kernel', r['with_mask_coordinate']['roofline']['kernel_ms'])
"
done
unset WBX_LIBRARY_PATH
bash tools/pmc_ens_binned.sh lat_fastest 2>&1 | grep "FETCH_SIZE" | head -2
