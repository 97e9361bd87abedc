#!/bin/bash
# A/B of the atom kernel's non-temporal hint (WBX_ATOMS_NT) on one box, with the FETCH_SIZE of each variant
mkdir -p gpurun_out
for v in "1 1" "1 0"; do
  set -- $v
  WBX_ATOMS_NT=$2 timeout 300 python tools/kbench_binned_ab.py "nt=$2" 2>gpurun_out/binned_nt_$1_$2.err | tee -a gpurun_out/binned_nt.jsonl
  tail -2 gpurun_out/binned_nt_$1_$2.err
done
for v in "1 0" "1 1"; do
  set -- $v
  ( WBX_ATOMS_NT=$2 bash tools/pmc_binned.sh lat_fastest | grep -v rocprofv3 | grep FETCH_SIZE | grep det_atoms ) 2>&1 | sed "s/^/nt=$2 /" | tee -a gpurun_out/binned_nt_fetch.txt
done
