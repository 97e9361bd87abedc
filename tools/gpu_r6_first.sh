#!/bin/bash
# Round 6, first GPU call: the slab-cache tests on the device, the box's host memory, det_atoms counters, a baseline bench line.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
{ free -g; nproc; } > gpurun_out/r6_box.txt 2>&1
timeout 600 python -m pytest tests/test_climatology_cache.py -m gpu -x -q > gpurun_out/r6_clim_cache_tests.txt 2>&1
tail -5 gpurun_out/r6_clim_cache_tests.txt
bash tools/pmc_binned.sh lon_fastest > gpurun_out/r6_pmc_det_atoms_lon.txt 2>&1
bash tools/pmc_binned.sh lat_fastest > gpurun_out/r6_pmc_det_atoms_lat.txt 2>&1
tail -12 gpurun_out/r6_pmc_det_atoms_lon.txt
cd $REPO
timeout 600 python bench.py > gpurun_out/r6_bench_base.json 2> gpurun_out/r6_bench_base.err
tail -c 1500 gpurun_out/r6_bench_base.json
