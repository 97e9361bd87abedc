#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_5.log
tail -3 gpurun_out/pytest_5.log
timeout 300 python tools/kbench_binned_ab.py "atoms,fullflush,dpp" 2>gpurun_out/binned_5.err | tee gpurun_out/binned_5.json
for lay in lon_fastest; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_binned_$lay -o r1 -- python $R/tools/kbench_binned.py $lay 6 > /dev/null 2>&1 )
done
python profiles/summarize_rocpd.py gpurun_out/trace_binned_*/r1_results.db 2>&1 | grep "wbx::" | head -8
rm -rf gpurun_out/trace_binned_lon_fastest
