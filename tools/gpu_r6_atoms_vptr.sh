#!/bin/bash
# Same-box A/B of det_atoms_kernel, round 6: per-lane pointers stepped by one vector add + execz skips + no `if (ok)` wrapper (the
# library) against the round-5 loop (make ab-novptr); then the binned parity tests on the library.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; O=$R/gpurun_out/r6_atoms_vptr.jsonl; : > $O
cd $R
for v in "" novptr "" novptr "" novptr; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_binned_ab.py "${v:-r6}" 2>/dev/null | tee -a $O
done
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_binned_ragged.py tests/test_gpu_cabi.py tests/test_replay.py tests/test_gpu_round3.py -m gpu -x -q > gpurun_out/r6_tests4.txt 2>&1
tail -5 gpurun_out/r6_tests4.txt
