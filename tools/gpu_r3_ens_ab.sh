#!/bin/bash
# Round 3 A/B on one box: library variants (make ab-*) x {pipelined LDS-DMA kernel, s1_xr / s1_xf1 kernel} x {lon, lat}-fastest
# usage: gpu_r3_ens_ab.sh "default stats64 slp ..." "1 0"
cd "$(dirname "$0")/.."
libs=${1:-"default stats64"}
pipes=${2:-"1 0"}
for round in 1 2; do
  for tag in $libs; do
    if [ "$tag" = default ]; then unset WBX_LIBRARY_PATH; else export WBX_LIBRARY_PATH=$PWD/weatherbenchx_amd/libwbx_hip_$tag.so; fi
    for pipe in $pipes; do
      export WBX_ENS_PIPE=$pipe
      python tools/kbench.py ens 2>&1 | grep "block= 64" | grep "sort" | sed "s#^#$tag pipe=$pipe: #"
      for layout in lon_fastest lat_fastest; do
        python bench.py --legs main --no-cpu --no-config5 --layout $layout 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$tag pipe=$pipe $layout main ms_per_step', round(r['ms_per_step'], 4), 'kernel_ms', r['roofline']['kernel_ms'], 'frac', r['roofline']['frac'])"
      done
    done
  done
done
