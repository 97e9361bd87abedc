#!/bin/bash
# A/B of the binned kernels on one box: slot kernel vs atom kernel, then the gpu tests
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_ab.log
tail -3 gpurun_out/pytest_ab.log
for v in "0 4" "1 4"; do
  set -- $v
  WBX_BINNED_ATOMS=$1 timeout 300 python tools/kbench_binned_ab.py "atoms=$1,pd=$2" 2>gpurun_out/binned_ab_$1_$2.err | tee -a gpurun_out/binned_ab.jsonl
  tail -2 gpurun_out/binned_ab_$1_$2.err
done
