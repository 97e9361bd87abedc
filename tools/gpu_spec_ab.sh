#!/bin/bash
# A/B of the 1440-point spectrum kernel's launch shape on one box, then its SQ / LDS counters
mkdir -p gpurun_out
for v in "12 1" "12 2" "12 3" "10 1" "8 1" "8 2" "6 2"; do
  set -- $v
  WBX_SPECTRUM_1440_TEAMS=$1 WBX_SPECTRUM_ROUNDS=$2 timeout 300 python bench.py --legs spectrum --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['spectrum']; print('teams=$1 rounds=$2', round(d['ms_per_step'],4), d['frac_of_hbm_peak'])" | tee -a gpurun_out/spec_ab.txt
done
bash tools/pmc_spectrum.sh > gpurun_out/pmc_spec_1440.txt 2>&1
cat gpurun_out/pmc_spec_1440.txt
