#!/bin/bash
# rocprofv3 kernel trace of tools/bench_ens_binned.py (one launch stream): per-kernel durations
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export WBX_ALTERNATE_STREAMS=0
for LAYOUT in "$@"; do
OUT=$REPO/gpurun_out/trace_ens_binned_$LAYOUT
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace -d $OUT/a -o t --output-format csv -- python $REPO/tools/bench_ens_binned.py $LAYOUT > $OUT/a.log 2>&1
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$OUT/a/*kernel_trace.csv')):
  dur = collections.defaultdict(list)
  for row in csv.DictReader(open(f)):
    dur[row['Kernel_Name'][:60]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
  for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v = sorted(v)
    print('$LAYOUT', k, 'n', len(v), 'avg_us', round(sum(v) / len(v), 1), 'median_us', round(v[len(v) // 2], 1))
PY
done
