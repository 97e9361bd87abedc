#!/bin/bash
# Where does the host spend a public-benchmark chunk?  (cProfile inflates everything ~2-3 x; the split is what matters.)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
LEG=${1:-public_chunk}
python -m cProfile -o /tmp/host.prof bench.py --legs $LEG --no-cpu --no-config5 --steps 300 --warmup 20 > /tmp/host.json 2>/dev/null
python - <<PY
import pstats, json
r = json.loads([l for l in open('/tmp/host.json').read().split('\n') if l.startswith('{')][-1])
leg = r.get('$LEG', r)
print('under cProfile:', {k: leg[k] for k in ('ms_per_chunk', 'ms_per_step') if k in leg})
st = pstats.Stats('/tmp/host.prof'); st.sort_stats('tottime').print_stats(45)
PY
python bench.py --legs $LEG --no-cpu --no-config5 --steps 300 --warmup 20 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); leg = r.get('$LEG', r)
print('without profiler:', {k: leg[k] for k in ('ms_per_chunk', 'ms_per_step') if k in leg}, leg.get('roofline', {}).get('kernel_ms'))"
