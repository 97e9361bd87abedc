#!/bin/bash
# one stream vs two alternating streams for the ensemble / indicator launches of a pipelined loop (engine._launch_context)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for alt in 1 0; do
    export WBX_ALTERNATE_STREAMS=$alt
    python bench.py --legs main,ensemble,config5 --no-cpu --config5-inits 60 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('alt=$alt main ms_per_step', round(r['ms_per_step'], 4), 'kernel_ms', r['roofline']['kernel_ms'], '| ensemble ms_per_step', round(r['ensemble']['ms_per_step'], 4), 'kernel', r['ensemble']['roofline']['kernel_ms'], '| config5 ms_per_chunk', round(r['config5']['ms_per_chunk'], 4), r['config5']['ms_per_chunk_by_pass_rank0'])"
    python tools/bench_ens_regions.py 2>/dev/null | tail -3 | sed "s/^/alt=$alt /"
  done
done
