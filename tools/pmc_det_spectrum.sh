#!/bin/bash
# Counters of zspec1440_det_kernel (wbx_det_spectrum) on a configs[4] z chunk: what bounds the fused det + spectra sweep?
# Each pass is a separate, bounded rocprofv3 --pmc run (--kernel-trace the only trace domain).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_det_spectrum
rm -rf $OUT; mkdir -p $OUT
run() { timeout 250 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$PASS -o pmc --output-format csv -- python $REPO/tools/kbench_det_spectrum.py > $OUT/$PASS.log 2>&1; }
PASS=a run FETCH_SIZE
PASS=b run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY
PASS=c run SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH
PASS=d run TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
PASS=e run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$OUT/*/*counter_collection.csv')):
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for row in csv.DictReader(open(f)):
    agg[row['Kernel_Name'][:52]][row['Counter_Name']].append(float(row['Counter_Value']))
  for k, c in agg.items():
    if 'zspec1440' in k:
      print(f.split('/')[-3], k, {n: (len(v), round(sum(v) / len(v), 1)) for n, v in c.items()})
for f in sorted(glob.glob('$OUT/a/*kernel_trace.csv')):
  dur = collections.defaultdict(list)
  for row in csv.DictReader(open(f)):
    dur[row['Kernel_Name'][:60]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
  for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v = sorted(v)
    print('trace', k, 'n', len(v), 'avg_us', round(sum(v) / len(v), 1), 'median_us', round(v[len(v) // 2], 1))
PY
