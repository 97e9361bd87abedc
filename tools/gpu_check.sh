#!/bin/bash
# One GPU-box visit: gpu tests, a small and a full bench run, the bare-shell --gpus 2 behaviour.  Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-a}
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_$TAG.log
( timeout 300 python bench.py --small --steps 2 --warmup 1 > gpurun_out/bench_small_$TAG.json ) 2> gpurun_out/bench_small_$TAG.err
echo "small rc=$?" >> gpurun_out/bench_small_$TAG.err
( timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.json ) 2> gpurun_out/bench_$TAG.err
echo "full rc=$?" >> gpurun_out/bench_$TAG.err
( timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_gpus2_$TAG.json ) 2> gpurun_out/bench_gpus2_$TAG.err
echo "gpus2 rc=$?" >> gpurun_out/bench_gpus2_$TAG.err
tail -5 gpurun_out/pytest_$TAG.log
tail -3 gpurun_out/bench_small_$TAG.err gpurun_out/bench_$TAG.err gpurun_out/bench_gpus2_$TAG.err
head -c 600 gpurun_out/bench_$TAG.json
