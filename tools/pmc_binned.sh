#!/bin/bash
# Counters of the fused binned kernel (wbx_det_binned) on one public-benchmark chunk; each pass is a separate, bounded
# rocprofv3 --pmc run.  FETCH_SIZE: 32-byte units, x2 on gfx950 (see profiles/r01_pmc_traffic.json).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
LAYOUT=${1:-lat_fastest}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_binned_$LAYOUT
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/a -o pmc --output-format csv -- python $REPO/tools/kbench_binned.py $LAYOUT 3 > $OUT/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --kernel-trace -d $OUT/b -o pmc --output-format csv -- python $REPO/tools/kbench_binned.py $LAYOUT 3 > $OUT/b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/c -o pmc --output-format csv -- python $REPO/tools/kbench_binned.py $LAYOUT 3 > $OUT/c.log 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$OUT/*/*counter_collection.csv')):
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for row in csv.DictReader(open(f)):
    agg[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
  for k, c in agg.items():
    if 'binned_kernel' in k or 'atoms_kernel' in k or 'atoms4_kernel' in k:
      print(k, {n: (len(v), sum(v) / len(v)) for n, v in c.items()})
PY
# memory-side counters (their own pass): texture-address / data busy, L2 hit rate, stalls
cd /tmp
timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace -d $OUT/d -o pmc --output-format csv -- python $REPO/tools/kbench_binned.py $LAYOUT 3 > $OUT/d.log 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$OUT/d/*counter_collection.csv')):
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for row in csv.DictReader(open(f)):
    agg[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
  for k, c in agg.items():
    if 'binned_kernel' in k or 'atoms_kernel' in k or 'atoms4_kernel' in k:
      print(k, {n: (len(v), sum(v) / len(v)) for n, v in c.items()})
PY
tail -3 $OUT/d.log
