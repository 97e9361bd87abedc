"""Turns what tools/profile_round.sh brings back (gpurun_out/prof/) into the tracked files of one round:
  rNN_bench_n1.json, rNN_bench_n1_lat_fastest.json   the bench lines, no profiler attached
  rNN_rocprofv3_summary.txt                          per-run kernel tables (rocprofv3 --kernel-trace) + counter tables
  rNN_kernel_stats.csv                               the kernel-trace tables as one CSV (run, kernel, calls, ...)
  rNN_pmc_traffic.json                               HBM bytes per launch per kernel (FETCH_SIZE x2 + WRITE_SIZE), keyed
                                                     by the names bench.py uses for its roofline entries
usage: python profiles/make_round_files.py gpurun_out/prof r01 [--traffic-only]"""
import csv
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
out = lambda name: os.path.join(here, f'{tag}_{name}')
def write_tables():
  shutil.copy(os.path.join(src, 'bench_n1.json'), out('bench_n1.json'))
  shutil.copy(os.path.join(src, 'bench_n1_lat_fastest.json'), out('bench_n1_lat_fastest.json'))
  text = open(os.path.join(src, 'summary.txt')).read()
  text = re.sub(r'== \S*/prof/', '== ', text)  # scratch path of the GPU box
  open(out('rocprofv3_summary.txt'), 'w').write(text)

  rows, run = [], None
  for line in text.split('\n'):
    if line.startswith('== '):
      run = line[3:].split('/')[0]
      continue
    m = re.match(r'^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$', line)
    if m and run and run.startswith('trace_'):
      rows.append([run] + list(m.groups()))
  with open(out('kernel_stats.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['run', 'kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'median_us'])
    w.writerows(rows)
  compare_events(rows)



def roofline_entries(node, path=''):
  """(where, roofline dict) for every roofline object of a bench.py line."""
  if isinstance(node, dict):
    if 'kernel' in node and 'kernel_ms' in node:
      yield path, node
    for k, v in node.items():
      yield from roofline_entries(v, f'{path}.{k}' if path else k)


def compare_events(rows):
  """rNN_events_vs_rocprof.txt: every traced bench.py process printed its own line, so the HIP-event duration of a kernel
  and rocprofv3's durations of the same launches come from ONE process."""
  lines = [f'{"traced run":20s} {"roofline entry":38s} {"kernel":26s} {"HIP events ms":>13s} {"rocprof median":>14s} {"avg":>9s} {"min":>9s} '
           f'{"calls":>6s} {"events/median":>13s}']
  for run in sorted({r[0] for r in rows}):
    path = os.path.join(src, run + '.json')
    if not os.path.exists(path):
      continue
    text = [l for l in open(path).read().split('\n') if l.startswith('{')]
    if not text:
      continue
    for where, roof in roofline_entries(json.loads(text[-1])):
      if where.startswith('lat_fastest'):
        continue
      token = re.split(r'[<( ]', roof['kernel'])[0]
      if token == 'wbx_det_binned':
        token = 'det_atoms_kernel'
      pair = 'true, 1>' if 'PAIRWISE' in roof['kernel'] else None
      skipna = 'SKIPNA_SORT' in roof['kernel']  # (template argument 97: another instantiation of the same kernel name)
      match = [r for r in rows if r[0] == run and ('wbx::' + token + '<' in r[1] or 'wbx::' + token + '(' in r[1])
               and (pair is None or pair in r[1]) and (pair is not None or 'true, 1>' not in r[1])
               and skipna == bool(re.search(r'true, 97[,>]', r[1]))]
      if not match:
        continue
      r = max(match, key=lambda r: float(r[3]))
      med, avg, mn = float(r[8]) / 1e3, float(r[4]) / 1e3, float(r[5]) / 1e3
      lines.append(f'{run[6:]:20s} {(where or "(main line)")[:38]:38s} {token[:26]:26s} {roof["kernel_ms"]:13.4f} {med:14.4f} {avg:9.4f} {mn:9.4f} '
                   f'{r[2]:>6s} {roof["kernel_ms"] / med:13.3f}')
  lines.append('')
  lines.append('HIP events: bench.py (main line: mean over the launches of the timed region, wbx_mark; side legs: N back-to-back launches per '
               'event pair in a separate pass).  rocprofv3 --kernel-trace: all launches of that kernel in the process, warm-up and '
               'first launches included -- hence the median.  The public-chunk entry of bench.py also holds the memset and the finish kernel.')
  open(out('events_vs_rocprof.txt'), 'w').write('\n'.join(lines) + '\n')


def bench_key(name: str) -> str:
  """rocprofv3's demangled name -> the label bench.py prints (template arguments spelled out)."""
  if '@' in name:  # launch-size variant of a kernel (tools/profile_round.sh): the suffix rides along
    base, suffix = name.rsplit('@', 1)
    return bench_key(base) + '@' + suffix
  n = name.replace('wbx::', '')
  if n.startswith('ens_atoms_kernel<51'):
    return 'ens_atoms_kernel'
  if n.startswith('ens_pipe_kernel<51, true, 97'):  # WBX_ENS_SKIPNA_SORT: skipna_ensemble's per-point member counts
    return 'ens_pipe_kernel_skipna'
  if n.startswith('ens_pipe_kernel<51'):
    return 'ens_pipe_kernel'
  if n.startswith('s1_xf1_kernel<EnsOpF32<51'):
    return 's1_xf1_kernel'
  if n.startswith('zspec1440_det_latfast_kernel'):
    return 'zspec1440_det_latfast_kernel'
  if n.startswith('zspec1440_det_kernel'):
    return 'zspec1440_det_kernel'
  n = re.sub(r'DetOp<float, 1, \d>', 'DetOp<float,DET6>', n)
  n = re.sub(r'EnsOpF32<51, true, 0>', 'EnsOpF32<51,true,SORT>', n)
  n = re.sub(r'EnsOpF32<51, true, 1>', 'EnsOpF32<51,true,PAIRWISE>', n)
  m = re.match(r's1_xr_kernel<(.*?), (\d), (?:false|true)>', n)
  if m:
    return f's1_xr_kernel<{m.group(1)},{m.group(2)}>'
  m = re.match(r's1_xf_kernel<(.*?) ?>$', n)
  if m:
    return f's1_xf_kernel<{m.group(1)}>'
  m = re.match(r'det_atoms_kernel<float, 1, (\d), (\d+), (\d)(?:, (?:true|false))*>', n)
  if m:
    return f'det_atoms_kernel<float,DET6,MM={m.group(1)},PD={m.group(2)},WM={m.group(3)}>'
  m = re.match(r'det_binned_kernel<float, 1, (\d), (\d+), (\d), (\d)>', n)
  if m:
    return f'det_binned_kernel<float,DET6,MM={m.group(1)},K={m.group(2)},PD={m.group(3)},WM={m.group(4)}>'
  if n.startswith('zspec1440_latfast_kernel'):
    return 'zspec1440_latfast_kernel'
  if n.startswith('zspec1440_kernel'):
    return 'zspec1440_kernel'
  m = re.match(r'zspec_fused_kernel<(\d), (\d+), (\d)>', n)
  if m:
    return f'zspec_fused_kernel<{m.group(1)},{m.group(2)},{m.group(3)}> (zonal spectrum, n=1440)'
  return n


def library_md5():
  import hashlib
  path = os.path.join(os.path.dirname(here), 'weatherbenchx_amd', 'libwbx_hip.so')
  return hashlib.md5(open(path, 'rb').read()).hexdigest() if os.path.exists(path) else None


def write_traffic():
  lib_md5 = library_md5()
  raw = json.load(open(os.path.join(src, 'pmc_raw.json')))
  traffic = {}
  for name, e in raw.items():
    if not isinstance(e, dict) or 'hbm_read_bytes' not in e:
      continue
    traffic[bench_key(name)] = dict(e, rocprof_name=name, library_md5=lib_md5)
  traffic['_library_md5'] = lib_md5  # bench.py drops a replayed traffic figure that was measured on another build
  traffic['_note'] = ('tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python '
                      'bench.py --steps 2 --warmup 1 --no-cpu` (+ --layout lat_fastest, + tools/kbench_binned.py for the '
                      'public-benchmark chunk); FETCH_SIZE is KiB and doubled per /opt/skills/guides/MI355X_MICROARCH.md '
                      '(gfx950 reports half of wide coalesced reads)')
  json.dump(traffic, open(out('pmc_traffic.json'), 'w'), indent=1)


write_traffic()
if '--traffic-only' not in sys.argv[3:]:  # tools/profile_round.sh writes the traffic file first: bench.py replays it
  write_tables()
  for extra in ('read_stream.json', 'pmc_ens.txt', 'pmc_binned_lon_fastest.txt', 'pmc_binned_lat_fastest.txt', 'pmc_spectrum.txt', 'pmc_spectrum_lat_fastest.txt',
                'spectrum_phase_profile.txt', 'spectrum_raw.txt', 'ubench.txt', 'config5_host_split.txt', 'new_time_labels.txt', 'kbench_ens.txt'):
    if os.path.exists(os.path.join(src, extra)):
      shutil.copy(os.path.join(src, extra), out(extra))
  print('wrote', [f for f in sorted(os.listdir(here)) if f.startswith(tag + '_')])
