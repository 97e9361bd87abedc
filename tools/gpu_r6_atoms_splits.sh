#!/bin/bash
# det_atoms_kernel: row splits per (cell, x tile) column on the public chunk (WBX_BINNED_TARGET_WAVES -> 1 .. 11 splits of the
# 721 rows; profiles/r06_det_atoms_ab.txt).  3588 columns on 4096 wave slots.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; O=$R/gpurun_out/r6_atoms_splits.jsonl; : > $O
cd $R
for rep in 1 2; do
for t in 3000 4096 8192 14000 17000 21000 25000 28000 32000 35800 39000; do
  WBX_KBENCH_LAYOUTS=lon_fastest WBX_BINNED_TARGET_WAVES=$t timeout 300 python tools/kbench_binned_ab.py "target$t" 2>/dev/null | tee -a $O
done
done
