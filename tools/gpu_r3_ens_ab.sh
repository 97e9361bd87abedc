#!/bin/bash
# Round 3 A/B on one box: {fp32 chain sums, fp64 sums} x {pipelined LDS-DMA kernel, s1_xr / s1_xf1 kernel} x {lon, lat}-fastest
cd "$(dirname "$0")/.."
for round in 1 2; do
  for lib in default weatherbenchx_amd/libwbx_hip_stats64.so; do
    if [ "$lib" = default ]; then unset WBX_LIBRARY_PATH; tag=stats32; else export WBX_LIBRARY_PATH=$PWD/$lib; tag=stats64; fi
    for pipe in 1 0; do
      export WBX_ENS_PIPE=$pipe
      python tools/kbench.py ens 2>&1 | grep "block= 64" | grep "sort" | sed "s#^#$tag pipe=$pipe: #"
      for layout in lon_fastest lat_fastest; do
        python bench.py --legs rmse_crps_37L --no-cpu --layout $layout 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
for k in ('rmse_crps_37L',):
  print('$tag pipe=$pipe $layout', k, 'ms_per_step', round(r[k]['ms_per_step'], 4), 'kernel_ms', r[k]['roofline']['kernel_ms'], 'frac', r[k]['roofline']['frac'])"
      done
    done
  done
done
