"""Merges a PARTIAL profile round (tools/profile_spectrum_update.sh: kernel traces of a few legs + the two bench lines) into
the tracked files of a round: the bench lines are replaced, and in rNN_rocprofv3_summary.txt / rNN_kernel_stats.csv /
rNN_events_vs_rocprof.txt the sections / rows / lines of the re-traced runs are replaced by the new ones, everything else
stays as the full round (tools/profile_round.sh + make_round_files.py) left it.
usage: python profiles/merge_partial_round.py gpurun_out/spec_update r03"""
import csv
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
out = lambda name: os.path.join(here, f'{tag}_{name}')


def sections(text):
  """{run: section text} of a summarize_rocpd.py listing, in order."""
  parts, run = {}, None
  for line in text.split('\n'):
    if line.startswith('== '):
      run = re.sub(r'^== (\S*/)?([^/\s]+)/r1_results\.db.*$', r'\2', line)
      parts[run] = ['== ' + run + '/r1_results.db']
    elif run is not None:
      parts[run].append(line)
  return {k: '\n'.join(v).rstrip('\n') + '\n' for k, v in parts.items()}


new = sections(open(os.path.join(src, 'summary.txt')).read())
old = sections(open(out('rocprofv3_summary.txt')).read())
old.update(new)
open(out('rocprofv3_summary.txt'), 'w').write('\n'.join(old[k] for k in sorted(old)))


def rows_of(run, text):
  rows = []
  for line in text.split('\n'):
    m = re.match(r'^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$', line)
    if m:
      rows.append([run] + list(m.groups()))
  return rows


fresh = {run: rows_of(run, text) for run, text in new.items() if run.startswith('trace_')}
kept = [r for r in csv.reader(open(out('kernel_stats.csv'))) if r and r[0] != 'run' and r[0] not in fresh]
rows = sorted(kept + [r for rs in fresh.values() for r in rs], key=lambda r: r[0])  # (stable: a run's rows keep their order)
with open(out('kernel_stats.csv'), 'w', newline='') as f:
  w = csv.writer(f)
  w.writerow(['run', 'kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'median_us'])
  w.writerows(rows)


def roofline_entries(node, path=''):
  if isinstance(node, dict):
    if 'kernel' in node and 'kernel_ms' in node:
      yield path, node
    for k, v in node.items():
      yield from roofline_entries(v, f'{path}.{k}' if path else k)


lines = open(out('events_vs_rocprof.txt')).read().split('\n')
head, body, tail = lines[0], [l for l in lines[1:] if l and not l.startswith('HIP events')], [l for l in lines if l.startswith('HIP events')]
body = [l for l in body if 'trace_' + l.split()[0] not in fresh]
for run, rs in fresh.items():
  path = os.path.join(src, run + '.json')
  text = [l for l in open(path).read().split('\n') if l.startswith('{')] if os.path.exists(path) else []
  if not text:
    continue
  for where, roof in roofline_entries(json.loads(text[-1])):
    if where.startswith('lat_fastest'):
      continue
    token = re.split(r'[<( ]', roof['kernel'])[0]
    match = [r for r in rs if 'wbx::' + token + '<' in r[1] or 'wbx::' + token + '(' in r[1]]
    if not match:
      continue
    r = max(match, key=lambda r: float(r[3]))
    med, avg, mn = float(r[8]) / 1e3, float(r[4]) / 1e3, float(r[5]) / 1e3
    body.append(f'{run[6:]:20s} {(where or "(main line)")[:38]:38s} {token[:26]:26s} {roof["kernel_ms"]:13.4f} {med:14.4f} {avg:9.4f} {mn:9.4f} '
                f'{r[2]:>6s} {roof["kernel_ms"] / med:13.3f}')
open(out('events_vs_rocprof.txt'), 'w').write('\n'.join([head] + sorted(body) + [''] + tail) + '\n')

for name in ('bench_n1.json', 'bench_n1_lat_fastest.json'):
  if os.path.exists(os.path.join(src, name)):
    shutil.copy(os.path.join(src, name), out(name))
print('merged', sorted(fresh), 'into', tag)
