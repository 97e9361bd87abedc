#!/bin/bash
# round 4: skipna_ensemble through the register-resident rank form (WBX_ENS_SKIPNA_SORT): parity, then the bench sub-entry
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_metrics.py tests/test_golden.py tests/test_corners.py -m gpu -x -q -k "skipna" 2>&1 | tail -15
for g in 0 1; do
WBX_ENS_SKIPNA_GENERIC=$g timeout 300 python bench.py --legs ensemble --no-cpu --no-config5 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('generic=$g', json.dumps(r['ensemble']['skipna_ensemble']))
"
done
