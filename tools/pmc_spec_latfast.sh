#!/bin/bash
# FETCH_SIZE, SQ and texture-addresser / L2 counters of zspec1440_latfast_kernel (separate rocprofv3 --pmc passes, no other
# trace domain): 2 leads x 37 levels x 721 x 1440, latitude-fastest, slabs of one group listed together
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=/tmp/pmc_lf; rm -rf $O; mkdir -p $O; cd $R
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/a -o r -- python tools/kbench_spectrum_raw.py 2 lat_fastest sorted > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $O/b -o r -- python tools/kbench_spectrum_raw.py 2 lat_fastest sorted > /dev/null 2>&1
timeout 200 rocprofv3 --pmc TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace -d $O/c -o r -- python tools/kbench_spectrum_raw.py 2 lat_fastest sorted > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('/tmp/pmc_lf/*/r_results.db')):
    c = sqlite3.connect(db).cursor()
    for r in c.execute("select substr(kernel_name,1,40), counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%zspec%' group by 1,2"): print(db.split('/')[-2], r)
PY
