#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/binned_6.jsonl
for o in 0 1; do
  WBX_PATCH_ORDER=$o timeout 300 python tools/kbench_binned_ab.py "order=$o" 2>gpurun_out/binned_6_$o.err | tee -a gpurun_out/binned_6.jsonl
done
