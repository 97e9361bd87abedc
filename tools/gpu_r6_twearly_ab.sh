#!/bin/bash
# Same-box A/B of z14_pair<.., TW_EARLY> (twiddle and mirror reads asked for ahead of the arithmetic that precedes their use):
# the fused det + spectra sweep with (library) and without it (make ab-zdtwlate); the three-wave spectrum kernel without (library)
# and with it (make ab-z14twearly).  (profiles/r06_det_spectrum_twearly_ab.txt was taken when the library had it and the variant,
# `zdtwlate`, did not.)
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
Z=$R/gpurun_out/r6_twearly_ab.txt; : > $Z
for rep in 1 2; do
for v in "" zdtwlate; do
  echo "== det + spectra, ${v:-library}" | tee -a $Z
  WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so timeout 300 python tools/kbench_det_spectrum.py 2>/dev/null | grep -E "^fused|^folded|max rel" | tee -a $Z
done
for v in "" z14twearly; do
  echo "== spectrum, ${v:-library}" | tee -a $Z
  WBX_LIBRARY_PATH=$R/weatherbenchx_amd/libwbx_hip${v:+_$v}.so timeout 300 python tools/kbench_spectrum.py 2>/dev/null | tail -2 | tee -a $Z
done
done
