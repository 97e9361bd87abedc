#!/bin/bash
# Round 3: pipelined ensemble kernel (LDS-DMA prefetch, fp32 chain sums) -- correctness, then same-box A/B against s1_xr_kernel
# (WBX_ENS_PIPE=0) on both layouts.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for round in 1 2; do
  for pipe in 1 0; do
    export WBX_ENS_PIPE=$pipe
    python tools/kbench.py ens 2>&1 | grep "block= 64" | grep "sort\|loadonly" | sed "s#^#pipe=$pipe: #"
    for layout in lon_fastest lat_fastest; do
    python bench.py --legs rmse_crps_37L,ensemble --no-cpu --layout $layout 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
for k in ('rmse_crps_37L', 'ensemble'):
  print('pipe=$pipe $layout', k, 'ms_per_step', round(r[k]['ms_per_step'], 4), 'kernel_ms', r[k]['roofline']['kernel_ms'], 'frac', r[k]['roofline']['frac'], r[k]['roofline']['kernel'][:30], 'check', r[k]['check'])"
    done
  done
done
