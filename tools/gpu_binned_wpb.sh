#!/bin/bash
# A/B of the atom kernel's waves per block (adjacent x tiles in one block) and the non-temporal hint on one box
mkdir -p gpurun_out
for v in "1 1" "1 0" "4 1" "4 0"; do
  set -- $v
  WBX_ATOMS_WPB=$1 WBX_ATOMS_NT=$2 timeout 300 python tools/kbench_binned_ab.py "wpb=$1,nt=$2" 2>gpurun_out/binned_wpb_$1_$2.err | tee -a gpurun_out/binned_wpb.jsonl
  tail -2 gpurun_out/binned_wpb_$1_$2.err
done
for v in "1 0" "4 0" "4 1"; do
  set -- $v
  ( WBX_ATOMS_WPB=$1 WBX_ATOMS_NT=$2 bash tools/pmc_binned.sh lat_fastest | grep -v rocprofv3 | grep FETCH_SIZE | grep det_atoms ) 2>&1 | sed "s/^/wpb=$1 nt=$2 /" | tee -a gpurun_out/binned_wpb_fetch.txt
done
