#!/bin/bash
# Spectrum tests, then the latitude-fastest 1440-point kernel against the transpose route (WBX_SPECTRUM_LATFAST=0) on one box
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_spectra.py tests/test_gpu_cabi.py tests/test_gpu_round2.py -m gpu -x -q -k "spectr or zonal or 1440" 2>&1 | tail -15 ) > gpurun_out/pytest_spec.log; grep -E "passed|failed|error" gpurun_out/pytest_spec.log | tail -3
for lf in 1 0; do
  echo "latfast=$lf"; WBX_SPECTRUM_LATFAST=$lf timeout 200 python tools/kbench_spectrum.py 8 lat_fastest 2>&1 | grep -v amdgpu.ids
done
timeout 200 python tools/kbench_spectrum.py 8 lon_fastest 2>&1 | grep -v amdgpu.ids
