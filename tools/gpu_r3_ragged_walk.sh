#!/bin/bash
# What does the ACCESS PATTERN of latitude-fastest chunks allow?  The binned kernel's load + arithmetic skeleton
# (tools/ubench/column_walk.hip) on rows of 721 floats, next to the same skeleton on 1440-point rows.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; O=$R/gpurun_out/ragged_walk.txt
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DCW_RAGGED $R/tools/ubench/column_walk.hip -o /tmp/cw_r 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/column_walk.hip -o /tmp/cw 2>/dev/null
{ echo "== rows of 721 floats (latitude-fastest public chunk)"; timeout 100 /tmp/cw_r; echo "== rows of 1440 floats"; timeout 100 /tmp/cw | sed -n 7p; } | tee $O
