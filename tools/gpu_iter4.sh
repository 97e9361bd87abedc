#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_4.log
tail -3 gpurun_out/pytest_4.log
rm -f gpurun_out/binned_4.jsonl
for pd in 2 4 8; do
  WBX_ATOMS_PD=$pd timeout 300 python tools/kbench_binned_ab.py "atoms,pd=$pd" 2>gpurun_out/binned_4_$pd.err | tee -a gpurun_out/binned_4.jsonl
done
