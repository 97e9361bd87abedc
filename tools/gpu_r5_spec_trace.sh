cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lay in lon_fastest lat_fastest; do
O=$R/gpurun_out/trace_spec_$lay; rm -rf $O; mkdir -p $O
WBX_CHUNK_REPLAY=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/a -o t --output-format csv -- python $R/bench.py --legs spectrum --no-cpu --no-config5 --steps 10 --warmup 3 --layout $lay > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$O/a/*kernel_trace.csv')):
  dur = collections.defaultdict(list)
  for row in csv.DictReader(open(f)):
    dur[row['Kernel_Name'][:70]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
  for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v = sorted(v)
    print('$lay', k, 'n', len(v), 'avg_us', round(sum(v) / len(v), 1), 'median_us', round(v[len(v) // 2], 1))
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
  gap[a[2][:28] + ' -> ' + b[2][:28]].append((b[0] - a[1]) / 1e3)
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:6]:
  v = sorted(v)
  print('$lay gap', k, 'n', len(v), 'median_us', round(v[len(v) // 2], 2), 'p10', round(v[len(v) // 10], 2), 'p90', round(v[9 * len(v) // 10], 2))
PY
done
