#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_spectra.py tests/test_replay.py -m gpu -x -q > gpurun_out/r6_tests5.txt 2>&1
tail -4 gpurun_out/r6_tests5.txt
Z=$R/gpurun_out/r6_zd_adopt.txt; : > $Z
for v in "" zd8f0 "" zd8f0; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  echo "== ${v:-library (12 teams, fetch at 5)}" | tee -a $Z
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_det_spectrum.py 2>/dev/null | grep -E "fused|max rel" | tee -a $Z
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_det_spectrum.py 8 2>/dev/null | grep -E "fused" | tee -a $Z
done
for v in "" zd8f0; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  echo "== bench config5,spectrum ${v:-library}" | tee -a $Z
  WBX_LIBRARY_PATH=$lib timeout 600 python bench.py --legs config5,spectrum --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d['legs'].items() if k in ('spectrum.det','config5.hits','config5')})" | tee -a $Z
done
