"""bench.py's output contract: ONE compact JSON line (the driver could not parse round 4's 20.9 KB line -- BENCH_r04.json
`parsed: null`), everything else in bench_full.json."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline')


def _inflate(result):
  """A full-size result with every leg bench.py can produce today plus what a later round might add: the nan-mask legs and
  long kernel names."""
  r = json.loads(json.dumps(result))
  for host in (r, r.get('lat_fastest', {})):
    pce = host.get('public_chunk_ens')
    if pce:
      pce['with_nan_mask'] = json.loads(json.dumps(pce.get('with_mask_coordinate', pce)))
      pce['with_skipna'] = json.loads(json.dumps(pce.get('with_mask_coordinate', pce)))
  r['roofline']['kernel'] = r['roofline']['kernel'] + ' ' + 'x' * 400
  r['config']['workload'] = r['config']['workload'] + ' ' + 'y' * 800
  return r


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0[3-9]_bench_n1*.json'))))
def test_compact_line_of_recorded_full_results(path):
  import bench
  full = json.load(open(path))
  for result in (full, _inflate(full)):
    text = bench.compact_line(result, 'bench_full.json')
    assert '\n' not in text and len(text.encode()) < 8192
    assert len(text.encode()) <= bench.LINE_LIMIT + 1500  # (the shedding order keeps it near the target)
    line = json.loads(text)
    for k in REQUIRED:
      assert k in line, k
    assert line['value'] == full['value'] and line['ms_per_step'] == full['ms_per_step']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel_ms', 'algorithmic_bytes_per_launch'):
      assert line['roofline'][k] == full['roofline'][k], k
    assert set(line['config']) >= {'workload', 'points_per_step_per_gpu', 'metrics'}
    if 'cpu_baseline' in full:
      cb = line['cpu_baseline']
      assert cb['value'] == full['cpu_baseline']['value'] and cb['cores'] == 1 and cb['kind'] == 'port' and cb['sample']
      assert cb['all_cores']['cores'] == full['cpu_baseline']['ensemble']['all_cores']['cores']
    legs = line['legs']
    assert 'configs1' in legs and abs(legs['configs1']['frac'] - full['configs1']['roofline']['frac']) < 1e-3
    assert all(len(json.dumps(v)) < 260 for v in legs.values())
    assert line['full'] == 'bench_full.json'


def test_emit_writes_the_full_result_beside_the_script(tmp_path, monkeypatch, capfd):
  import bench
  monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
  monkeypatch.setattr(bench, '_JSON_FD', None)
  full = json.load(open(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0*_bench_n1.json')))[-1]))
  bench._emit(full)
  out, err = capfd.readouterr()
  lines = [l for l in out.splitlines() if l.strip()]
  assert len(lines) == 1 and len(lines[0]) < 8192
  assert json.loads(lines[0])['full'] == 'bench_full.json'
  assert json.load(open(tmp_path / 'bench_full.json')) == full
  assert json.loads(err.strip().splitlines()[-1]) == full


def test_skipped_line_is_one_small_json_line():
  """`--gpus 8` on a box without 8 devices: one JSON line with "skipped" and exit code 0 (the dry run of the N > 1 launch path)."""
  proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'],
                        capture_output=True, text=True, timeout=300, env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
  assert proc.returncode == 0, proc.stderr[-2000:]
  lines = [l for l in proc.stdout.splitlines() if l.strip()]
  assert len(lines) == 1
  line = json.loads(lines[0])
  assert line['skipped'] is True and line['n_gpus'] == 8 and line['value'] is None


@pytest.mark.gpu
def test_small_bench_prints_one_parsable_line():
  proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--small', '--steps', '2', '--warmup', '1', '--prewarm-ms', '0',
                         '--cpu-workers', '2'], capture_output=True, text=True, timeout=900)
  assert proc.returncode == 0, proc.stderr[-4000:]
  lines = [l for l in proc.stdout.splitlines() if l.strip()]
  assert len(lines) == 1 and len(lines[0]) < 8192, (len(lines), len(lines[0]) if lines else 0)
  line = json.loads(lines[0])
  for k in REQUIRED + ('cpu_baseline', 'legs'):
    assert k in line, k
  assert line['roofline']['frac'] > 0 and line['cpu_baseline']['value'] > 0
  assert json.load(open(os.path.join(ROOT, line['full'])))['value'] == line['value']


def test_gpus_8_launch_command_is_the_drivers(monkeypatch):
  """`python bench.py --gpus 8` from a bare shell on an 8-GPU node re-launches itself exactly as the driver would:
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...`,
  one rank per GPU, with HSA_ENABLE_IPC_MODE_LEGACY=0 in the ranks' environment (dmabuf IPC for RCCL)."""
  import argparse
  import bench
  from weatherbenchx_amd import _hip
  seen = {}
  monkeypatch.setattr(_hip, 'device_count', lambda: 8)
  monkeypatch.setattr(bench.os.path, 'exists', lambda p: True)
  monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
  monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '7', '--warmup', '2'])
  monkeypatch.delenv('HSA_ENABLE_IPC_MODE_LEGACY', raising=False)
  rc = bench.self_launch(argparse.Namespace(gpus=8, steps=7, warmup=2, backend='nccl'))
  assert rc == 0
  cmd = seen['cmd']
  assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
  assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
  script = cmd.index(os.path.join(ROOT, 'bench.py'))
  assert cmd[script + 1:] == ['--gpus', '8', '--steps', '7', '--warmup', '2']
  assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_n8_line_names_the_librarys_collective():
  """At N > 1 on RCCL ranks the payload collective is the library's (wbx_comm_create / wbx_acc_allreduce) unless
  WBX_COLLECTIVE=torch, and the line says so: backend, ranks, the collective's bytes and its own time."""
  import bench
  assert bench.pick_collective(1, 'nccl', {}) is None
  assert bench.pick_collective(8, 'nccl', {}) == bench.COLLECTIVE_CABI
  assert bench.pick_collective(8, 'nccl', {'WBX_COLLECTIVE': 'torch'}) == 'torch.distributed nccl'
  assert bench.pick_collective(2, 'gloo', {}) == 'torch.distributed gloo'

  class _Comm:
    nranks = 8
    timings = {'collectives': 21, 'us_total': 21 * 35.0, 'us_last': 33.3, 'bytes_last': 17146728}

  class _Env:
    comm = _Comm()
  rec = bench._collective_record(_Env())
  assert rec == {'rccl_ranks': 8, 'collectives': 21, 'bytes': 17146728, 'us_last': 33.3, 'us_mean': 35.0}
  result = {'metric': bench.METRIC, 'value': 9.0e11, 'unit': 'evals/s', 'n_gpus': 8, 'steps': 20, 'warmup': 5, 'ms_per_step': 1.4,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32+f64', 'data': 'synthetic',
            'config': {'workload': 'w', 'rccl_ranks': 8, 'collectives_per_step': 1.0, 'collective_backend': bench.COLLECTIVE_CABI,
                       'collective': rec}}
  line = json.loads(bench.compact_line(result))
  assert line['config']['collective_backend'] == bench.COLLECTIVE_CABI and line['config']['collective']['bytes'] == 17146728
  assert line['config']['rccl_ranks'] == 8 and line['config']['collective']['us_mean'] == 35.0
