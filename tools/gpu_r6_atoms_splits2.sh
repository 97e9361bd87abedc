#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; O=$R/gpurun_out/r6_atoms_splits2.jsonl; : > $O
cd $R
for rep in 1 2 3; do
for t in 8192 4096 3000; do
  WBX_BINNED_TARGET_WAVES=$t timeout 300 python tools/kbench_binned_ab.py "target$t" 2>/dev/null | tee -a $O
done
done
