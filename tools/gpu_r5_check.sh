#!/bin/bash
# Round 5 checkpoint on the GPU box: the whole -m gpu suite, smoke(), the default bench line (what the driver runs).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r5}
( time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_$TAG.log 2>&1
tail -6 gpurun_out/pytest_$TAG.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 ) > gpurun_out/smoke_$TAG.log
tail -2 gpurun_out/smoke_$TAG.log
( time timeout 900 python bench.py > gpurun_out/bench_$TAG.json ) 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?" >> gpurun_out/bench_$TAG.err
tail -6 gpurun_out/bench_$TAG.err | cut -c1-300
wc -c gpurun_out/bench_$TAG.json
cat gpurun_out/bench_$TAG.json
cp bench_full.json gpurun_out/bench_full_$TAG.json 2>/dev/null
