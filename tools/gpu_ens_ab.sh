#!/bin/bash
# Same-box A/B of the ensemble kernel: the library as built against a second build (WBX_LIBRARY_PATH).  usage: gpu_ens_ab.sh <other.so>
cd "$(dirname "$0")/.."
other=${1:-weatherbenchx_amd/libwbx_hip_batcher.so}  # make -C weatherbenchx_amd/csrc ab-batcher
for round in 1 2; do
  for lib in default "$other"; do
    if [ "$lib" = default ]; then unset WBX_LIBRARY_PATH; else export WBX_LIBRARY_PATH=$PWD/$lib; fi
    python tools/kbench.py ens 2>&1 | grep "block= 64" | grep "sort\|loadonly" | sed "s#^#$lib: #"
    python bench.py --legs main,ensemble --no-cpu --no-config5 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
for k, leg in (("main", r), ("ensemble", r["ensemble"])):
  print('$lib', k, 'ms_per_step', round(leg['ms_per_step'], 4), 'kernel_ms', leg['roofline']['kernel_ms'], 'frac', leg['roofline']['frac'], 'check', leg['check'])"
  done
done
