#!/bin/bash
# round 4: the latitude-fastest fused det + spectra kernel: parity of the entry point, then the configs[3] composite (lat-fastest)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export WBX_FUSE_DET_SPECTRA_LATFAST=1
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "slabs or latitude_fastest_chunks" 2>&1 | tail -3
for env in "$@" ""; do
  [ -z "$env" ] && [ $# -gt 0 ] && break
  echo "== ${env:-default}"
  env $env timeout 300 python bench.py --legs spectrum --no-cpu --no-config5 --steps 10 --warmup 3 --layout lat_fastest 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().split('\n')[-1])
s = r.get('spectrum', r)
c = s['with_deterministic_suite']
print('spectrum ms/step', round(s['ms_per_step'], 4), 'frac', s['roofline']['frac'], '| composite ms/chunk', round(c['ms_per_chunk'], 4), 'kernel_ms', c.get('roofline', {}).get('kernel_ms'), 'frac', c.get('roofline', {}).get('frac'), c['launches_per_chunk'])
"
done
