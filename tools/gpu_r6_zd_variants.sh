#!/bin/bash
# Same-box A/B of the fused det + spectra sweep against variant libraries (profiles/r06_det_spectrum_*): make ab-<variant> first.
#   usage: bash tools/gpu_r6_zd_variants.sh zd12f5 zd12f45        (three waves per SIMD)
#          bash tools/gpu_r6_zd_variants.sh zdspread | zdf32 | zdk1 zdk2 zdk4 zdk6 zdk8 zdk16   (spread fetches, fp32 chains, knock-outs)
#          bash tools/gpu_r6_zd_variants.sh zdflat | zdf32p2 zdf32p3 | head   (flat loads; packed-fp32 lanes; `head` = a library built from
#                                                                               another commit, copied to weatherbenchx_amd/libwbx_hip_head.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
Z=$R/gpurun_out/r6_zd_variants.txt; : > $Z
for rep in 1 2; do
for v in "" "$@"; do
  lib=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so
  echo "== ${v:-library}" | tee -a $Z
  WBX_LIBRARY_PATH=$lib timeout 300 python tools/kbench_det_spectrum.py 2>/dev/null | grep -E "fused|folded|max rel" | tee -a $Z
done
done
