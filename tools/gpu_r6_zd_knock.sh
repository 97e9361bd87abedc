#!/bin/bash
# Timing diagnostics of the fused det + spectra sweep (WRONG results): zdk1 = every row re-reads the team's first row (no HBM
# stream), zdk2 = no deterministic lanes, zdk4 = no loads after the first row, zdk6 = both.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
Z=$R/gpurun_out/r6_zd_knock.txt; : > $Z
for v in "" zdk1 zdk2 zdk4 zdk6 "" zdk1 zdk2 zdk4 zdk6; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  echo "== ${v:-library}" | tee -a $Z
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_det_spectrum.py 2>/dev/null | grep -E "fused|spectra alone" | tee -a $Z
done
