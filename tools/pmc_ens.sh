#!/bin/bash
# SQ counters for the ensemble stage-1 kernel (separate rocprofv3 --pmc pass; never combined with other trace domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_ens; rm -rf $O; mkdir -p $O; cd $R
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/a -o r -- python tools/kbench.py ens > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $O/b -o r -- python tools/kbench.py ens > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob('gpurun_out/pmc_ens/*/r_results.db')):
    con = sqlite3.connect(db); c = con.cursor()
    rows = c.execute("select substr(kernel_name,1,60), counter_name, count(*), avg(value), avg(duration) from counters_collection where (kernel_name like '%ens_pipe_kernel<51%' or kernel_name like '%EnsOpF32<51%') and workgroup_size=64 group by 1,2").fetchall()
    for r in rows: print(db.split('/')[-2], r)
PY
