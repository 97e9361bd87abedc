#!/bin/bash
# Fused det + spectra sweep at three waves per SIMD (twelve teams per CU, 161-168 VGPRs: the next row's p, t, c asked for at the
# end of a pair instead of across the transform) against the library (eight teams, 214 VGPRs), same box, alternating.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
Z=$R/gpurun_out/r6_zd3waves.txt; : > $Z
for v in "" zd12f5 zd12f45 "" zd12f5 zd12f45; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  echo "== ${v:-library}" | tee -a $Z
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_det_spectrum.py 2>/dev/null | tee -a $Z
done
