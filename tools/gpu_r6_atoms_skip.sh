#!/bin/bash
# Same-box A/B of det_atoms_kernel: execz skips around an idle entry's six fp64 FMAs (the library) against the round-5 code
# (make ab-noskip).  Both layouts, smooth and random land masks; checksums must agree.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; O=$R/gpurun_out/r6_atoms_skip.jsonl; : > $O
cd $R
for v in "" noskip "" noskip; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_binned_ab.py "${v:-skip}" 2>/dev/null | tee -a $O
done
