#!/bin/bash
# round 4: the latitude-fastest fused det + spectra kernel: parity, then the configs[3] composite on both layouts
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "slabs or latitude_fastest_chunks" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_spectra.py -m gpu -x -q -k "spectr" 2>&1 | tail -5
for layout in lat_fastest lon_fastest; do
  timeout 300 python bench.py --legs spectrum --no-cpu --no-config5 --steps 10 --warmup 3 --layout $layout 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().split('\n')[-1])
s = r.get('spectrum', r)
c = s['with_deterministic_suite']
print('$layout', 'spectrum ms/step', round(s['ms_per_step'], 4), 'frac', s['roofline']['frac'], '| composite ms/chunk', round(c['ms_per_chunk'], 4), {k: v for k, v in c.items() if k not in ('workload', 'ms_per_chunk_runs', 'check')})
"
done
