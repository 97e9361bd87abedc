#!/bin/bash
# Round 5: where a replayed chunk's time goes on the GPU -- rocprofv3 kernel trace of the chunk loop with chunk records on:
# per chunk the kernels, their durations and the gaps between them (steady state: the last chunks of the loop).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
for CASE in "$@"; do
W=${CASE%%:*}; LAY=${CASE##*:}
OUT=$REPO/gpurun_out/trace_replay_${W}_$LAY
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace -d $OUT/a -o t --output-format csv -- python $REPO/tools/bench_replay.py $W $LAY n=40 > $OUT/a.log 2>&1
python - <<PY
import csv, glob
for f in sorted(glob.glob('$OUT/a/*kernel_trace.csv')):
  rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
  # the last replay run of the tool: 40 chunks with records on, before the event-marked pass -- take a window of kernels well inside it
  names = [n for _, _, n in rows]
  main = max(set(names), key=lambda n: sum(e - s for s, e, m in rows if m == n))
  idx = [i for i, (_, _, n) in enumerate(rows) if n == main]
  lo, hi = idx[-30], idx[-14]   # inside the last 40-chunk loop (the 12-chunk marked pass comes after it)
  prev_end = None
  print('$W $LAY  main kernel:', main[:70])
  for s, e, n in rows[lo:hi]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f'  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  {n[:80]}')
    prev_end = e if prev_end is None else max(prev_end, e)
  span = (rows[hi - 1][1] - rows[lo][0]) / 1e3
  nmain = sum(1 for s, e, n in rows[lo:hi] if n == main)
  busy = sum((e - s) for s, e, n in rows[lo:hi]) / 1e3
  print(f'  window: {nmain} chunks, {span / nmain:.1f} us per chunk, kernels busy {busy / nmain:.1f} us per chunk')
PY
done
