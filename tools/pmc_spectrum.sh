#!/bin/bash
# SQ / LDS counters for the fused zonal-spectrum kernel (separate rocprofv3 --pmc passes; never combined with other trace domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_spec; rm -rf $O; mkdir -p $O; cd $R
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/a -o r -- python tools/bench_spectrum.py 2 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/b -o r -- python tools/bench_spectrum.py 2 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS --kernel-trace -d $O/c -o r -- python tools/bench_spectrum.py 2 > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('gpurun_out/pmc_spec/*/r_results.db')):
    con = sqlite3.connect(db); c = con.cursor()
    try:
        rows = c.execute("select substr(kernel_name,1,40), counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%zspec%' group by 1,2").fetchall()
    except Exception as e:
        print(db, e); continue
    for r in rows: print(db.split('/')[-2], r)
PY
