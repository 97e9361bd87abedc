#!/bin/bash
# Phase profile of zspec1440_latfast_kernel (s_memtime stamps of wave 0 of every block, cycles per step of <= 12 row pairs):
# usage (GPU box): bash tools/spec_latfast_phase_profile.sh
rm -f /tmp/lf.txt; WBX_SPECTRUM_PROF=/tmp/lf.txt timeout 200 python tools/kbench_spectrum_raw.py 8 lat_fastest > /dev/null 2>&1; python - <<'PY'
rows=[list(map(int,l.split()[1:])) for l in open('/tmp/lf.txt') if l.startswith('latfast')]
import numpy as np
t=np.array(rows[2:],float).sum(0)
names=['steps','wait barrier A','stage + barrier B','issue loads','pass-1 LDS loads','passes + unpack']
for n,v in zip(names[1:],t[1:]/t[0]): print(f'{n:20s} {v:9.0f} cycles per step')
print('total', (t[1:]/t[0]).sum(), 'steps per launch (256 blocks):', t[0]/len(rows[2:])/256)
PY
