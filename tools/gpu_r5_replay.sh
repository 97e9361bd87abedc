#!/bin/bash
# Round 5: chunk records -- parity (replay on / off bit for bit) and host time per chunk on the public benchmark's chunks.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-a}
( timeout 1200 python -m pytest tests/test_replay.py tests/test_pipeline.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r5_pytest_replay_$TAG.log
tail -15 gpurun_out/r5_pytest_replay_$TAG.log
OUT=gpurun_out/r5_bench_replay_$TAG.txt
: > $OUT
for w in det ens ens_mask ens_nan spec; do
  for lay in lon_fastest lat_fastest; do
    timeout 300 python tools/bench_replay.py $w $lay >> $OUT 2>&1
  done
done
timeout 300 python tools/bench_replay.py ens ifs >> $OUT 2>&1
grep -v amdgpu.ids $OUT | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l[:300]); continue
    d=json.loads(l)
    print('%-8s %-11s 1-stream ord %s rep %s | 2-stream ord %s rep %s | kernel %.4f ratio %.3f %s %s'%(d['which'],d['layout'],d['ordinary_one_stream_ms_per_chunk'],d['replay_one_stream_ms_per_chunk'],d['ordinary_ms_per_chunk'],d['replay_ms_per_chunk'],d['kernel_ms_per_chunk'],d['replay_over_kernel'],d['stats'],d['refusals'][:1]))
"
