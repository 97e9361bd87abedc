#!/bin/bash
# What the driver runs at round end, on one box: the whole `-m gpu` suite, smoke(), the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/check_pytest_gpu.txt 2>&1; tail -4 gpurun_out/check_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/check_smoke.txt 2>&1; tail -2 gpurun_out/check_smoke.txt
timeout 900 python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; tail -c 2500 gpurun_out/check_bench.json
