export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_corners.py tests/test_metrics.py -m gpu -x -q -k "skipna" 2>&1 | tail -4
for v in 0 1 0 1; do
  WBX_ENS_PIPE_SKIPNA=$v timeout 300 python bench.py --legs ensemble --no-cpu --no-config5 --steps 10 --warmup 3 > /dev/null 2>/tmp/err.txt
  python - <<PY
import json
d=json.load(open('bench_full.json'))
s=d['ensemble']['skipna_ensemble']
print('WBX_ENS_PIPE_SKIPNA=$v', s['roofline']['kernel'][:70], s['roofline']['kernel_ms'], s['roofline']['frac'], s['crps'])
PY
done
