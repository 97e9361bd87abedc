#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests -m gpu -x -q -k "spectr" 2>&1 | tail -4 ) 
for mb in 16 32 64 128 256; do
  echo "tile_mb=$mb"; WBX_SPECTRUM_TILE_MB=$mb timeout 200 python tools/bench_spectrum.py 8 lat_fastest 2>&1 | grep -i "spectra" | tail -2
done
timeout 200 python tools/bench_spectrum.py 8 lon_fastest 2>&1 | grep -i "spectra" | tail -2
