#!/bin/bash
# The round's closing GPU-box visit on a small budget: gpu tests, smoke(), the default bench line.  Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-final}
( timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_$TAG.log
tail -3 gpurun_out/pytest_$TAG.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 ) > gpurun_out/smoke_$TAG.log
tail -2 gpurun_out/smoke_$TAG.log
( timeout 400 python bench.py > gpurun_out/bench_$TAG.json ) 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?" >> gpurun_out/bench_$TAG.err
tail -2 gpurun_out/bench_$TAG.err
head -c 400 gpurun_out/bench_$TAG.json
