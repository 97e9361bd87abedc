#!/bin/bash
# round 4: where the time of zspec1440_det_latfast_kernel goes (knock-out instantiations: wrong results, timing only; they exist
# in the diagnostic build only: make -C weatherbenchx_amd/csrc diag)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/libwbx_hip_diag.so WBX_FUSE_DET_SPECTRA_LATFAST=1
for k in ${KNOCKS:-0 1 2 3 4 8 11}; do
  WBX_ZL_KNOCK=$k timeout 300 python bench.py --legs spectrum --no-cpu --no-config5 --steps 4 --warmup 2 --layout lat_fastest 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().split('\n')[-1])
c = r.get('spectrum', r)['with_deterministic_suite']
print('knock $k: composite ms/chunk', round(c['ms_per_chunk'], 4), 'kernel_ms', c.get('roofline', {}).get('kernel_ms'))
"
done
