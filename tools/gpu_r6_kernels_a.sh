#!/bin/bash
# Round 6 kernel A/Bs on one box: det_atoms rows in flight (4 = library / 8 / 12), fused det + spectra with fp32 chains (library)
# against fp64 lanes (make ab-zdf64); then the GPU tests that cover both kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
O=$R/gpurun_out/r6_atoms_pd.jsonl; : > $O
for v in "" pd8 pd12 "" pd8 pd12; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_binned_ab.py "${v:-pd4}" 2>/dev/null | tee -a $O
done
Z=$R/gpurun_out/r6_zd_chains.txt; : > $Z
for v in "" zdf64 "" zdf64; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  echo "== ${v:-f32chains}" | tee -a $Z
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_det_spectrum.py 2>/dev/null | tee -a $Z
done
timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_spectra.py tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/r6_tests3.txt 2>&1
tail -15 gpurun_out/r6_tests3.txt
