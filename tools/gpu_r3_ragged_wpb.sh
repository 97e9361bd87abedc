#!/bin/bash
# Same-box A/B of the binned atom kernel on latitude-fastest chunks (721-point rows): 4 (the library) / 1 / 2 / 6 / 12 waves per
# block on adjacent x tiles, meeting at a barrier every 64 rows (make ab-wpb1 ab-wpb2 ab-wpb6 ab-wpb12).  Checksums must agree.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; O=$R/gpurun_out/ragged_wpb.jsonl; : > $O
cd $R
for v in ${WPB_VARIANTS:-wpb4 wpb1 wpb2 wpb6 wpb12 wpb4 wpb1}; do
  [ "$v" = wpb4 ] && v=""; lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  WBX_KBENCH_LAYOUTS=lat_fastest WBX_LIBRARY_PATH=$lib timeout 200 python tools/kbench_binned_ab.py "${v:-wpb4}" 2>/dev/null | tee -a $O
done
