#!/bin/bash
# SQ counters of the binned atom kernel (public chunk, longitude-fastest) next to its load + arithmetic skeleton
# (tools/ubench/column_walk.hip): where do the extra cycles of the real kernel go?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bvs; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/column_walk.hip -o /tmp/column_walk 2>/dev/null
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
C2="SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
C3="TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
i=0
for C in "$C1" "$C2" "$C3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $O/real$i -o r -- python $R/tools/kbench_binned.py lon_fastest 3 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $O/skel$i -o r -- /tmp/column_walk > /dev/null 2>&1
done
python - <<PY
import glob, sqlite3
for db in sorted(glob.glob('$O/*/r_results.db')):
  rows = sqlite3.connect(db).execute("select substr(kernel_name, 1, 58), counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%det_atoms_kernel%' or kernel_name like '%walk_kernel<1, 4, 4>%' or kernel_name like '%walk_kernel<1, 4, 8>%' group by 1, 2").fetchall()
  for k, c, n, v, d in rows: print(db.split('/')[-2], k, c, n, '%.0f' % v, 'dur_us %.1f' % (d / 1e3))
PY
rm -rf $O
