#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_loaders.py tests/test_climatology_cache.py tests/test_replay.py tests/test_abi.py -m gpu -x -q > gpurun_out/r6_tests2.txt 2>&1
tail -5 gpurun_out/r6_tests2.txt
timeout 300 python tools/bench_host_transpose.py > gpurun_out/r6_host_transpose.txt 2>&1
cat gpurun_out/r6_host_transpose.txt
timeout 600 python bench.py --small > gpurun_out/r6_bench_small.json 2> gpurun_out/r6_bench_small.err || tail -30 gpurun_out/r6_bench_small.err
tail -c 600 gpurun_out/r6_bench_small.json
timeout 900 python bench.py --legs config5,lat_fastest --no-cpu > gpurun_out/r6_bench_c5.json 2> gpurun_out/r6_bench_c5.err || tail -30 gpurun_out/r6_bench_c5.err
tail -c 1200 gpurun_out/r6_bench_c5.json
