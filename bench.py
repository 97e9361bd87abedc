#!/usr/bin/env python3
"""Benchmark of the scoring hot path on MI355X (driver contract: one JSON line on rank 0).

Output: stdout carries ONE compact JSON line (< 6 KB: headline + `roofline` + `cpu_baseline` + `legs`, a table of one short
record per side leg); the full result of every leg goes to `bench_full.json` beside this script (also `gpurun_out/`) and to
stderr.  `compact_line()` builds the line; tests/test_bench_line.py holds its size.

`python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment the script launches
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU,
backend 'nccl' = RCCL); under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  Fewer visible devices than N: one JSON
line with "skipped" and exit code 0.

Main line = the configuration BASELINE.json's metric and north_star name: area-weighted CRPS (fair) + unbiased spread/skill +
unbiased ensemble-mean RMSE + ensemble-mean RMSE on ONE forecast field float32[1 init, 37 level, 51 member, 721, 1440]
(7.99 GB) against float32[1, 37, 721, 1440] targets, reduced over (init_time, latitude, longitude) with GridAreaWeighting --
through the drop-in API (Statistic.compute -> Aggregator.aggregate_statistics -> metric_values), inputs resident in HBM.
A "step" is one such pass; the K timed steps are software-pipelined one deep like pipeline.evaluate_chunks (every step is
launched and finished inside the timed region).  With N ranks every rank owns one such field = one (init, lead) time slice
(weak scaling): the step's sums stay in HBM (engine.Accumulation) and are summed over the ranks with ONE all-reduce of the
device buffer per step (distributed.resolve_state; the layout is exchanged once, before the first step) -- on RCCL ranks through
the LIBRARY's communicator (wbx_comm_create / wbx_acc_allreduce; WBX_COLLECTIVE=torch for the A/B); `config.collective` carries
its bytes and its own time on the library's stream.

value    = (grid points per step x 4 metrics x N) / wall time per step (max over ranks), evals/s; ensemble members are not
           extra points
roofline = algorithmic bytes (208 B/point: 51 members + the target, read once) / mean duration of the ensemble kernel
           (HIP events on the launch stream, separate pass), against 8.0 TB/s

Side legs in the same line (N = 1 unless noted):
  configs1       BASELINE.json configs[1]: RMSE / MSE / MAE / bias / ACC / activity on f32[40 init, 10 lead, 5 level, 721, 1440]
                 p, t + (dayofyear, hour) climatology (the main line of rounds 1-2)
  ensemble       configs[2]: 6 variables x f32[8, 51, 721, 1440]; `pairwise_form`: the O(M^2) pair kernel alone
  public_chunk   the public benchmark's chunk (1 init x 12 lead x 13 level, 34 region x land/sea bins, masked)
  spectrum       configs[3]: zonal spectra of two f32[8, 37, 721, 1440] fields
  lat_fastest    the main line, configs1 and spectrum again on latitude-fastest arrays (the layout of the public archives)
  config5        configs[4] (every N): the full suite streamed as [1 init x 20 lead x 37 level] chunks through
                 pipeline.evaluate_passes, 366 daily inits in contiguous runs per rank, accumulators in HBM, ONE all-reduce
                 (the library's RCCL communicator) at the end; strong scaling (total work fixed).  The FIELDS come from a
                 resident pool (their H2D excluded, stated); the CLIMATOLOGY is a whole [366, 4, 37, 721, 1440] calendar in
                 host memory behind a 48-slab device pool (climatology_cache.py): 4 new (dayofyear, hour) slabs = 0.61 GB per
                 chunk cross PCIe one chunk ahead -- the leg's `ms` INCLUDES that stream and is bound by it;
                 `config5.hits` is the same loop with every slab already in the pool (kernels + host per chunk)
  lat.config5    the same evaluations fed from a LATITUDE-FASTEST archive (.npy files [.., longitude, latitude] in /dev/shm)
                 through the transposing loader and slab pool (`device_layout='lon_fastest'`): host cost of the transposition
                 against the plain copy, and the chunk loop on what arrived (longitude-fastest on the device)
  box            the shader clock this box sustains under load (wbx_clock_probe): boxes of the pool differ by 4-8 %
  cpu_baseline   the oracle's "reference structure" NumPy path (one pass per statistic + two einsums, float32
                 statistics; oracle/wbx_oracle.py) on bounded samples: the ensemble suite (the main line's workload) and the
                 deterministic suite (configs1), each on one process and on os.cpu_count() worker processes
                 (oracle/cpu_workers.py); rank 0, N = 1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
METRIC = 'grid-point·metric evals/s (0.25° 721×1440×37L) at 1/2/4/8 GPUs; % HBM roofline'  # BASELINE.json's string


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--main-streams', type=int, default=1, choices=(1, 2), help='HIP streams the steps of the main line are dealt to')
  ap.add_argument('--prewarm-ms', type=float, default=150.0,
                  help='untimed steps of the main loop before the W warm-up steps, until the device has been under load this long')
  ap.add_argument('--inits', type=int, default=40)
  ap.add_argument('--leads', type=int, default=10)
  ap.add_argument('--levels', type=int, default=5)
  ap.add_argument('--layout', choices=['lon_fastest', 'lat_fastest'], default='lon_fastest')
  ap.add_argument('--no-ens', action='store_true', help='skip the single-GPU side legs (configs1, ensemble, public chunk, spectrum, lat_fastest)')
  ap.add_argument('--ens-slices', type=int, default=8)
  ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
  ap.add_argument('--no-config5', action='store_true', help='skip the streamed full-suite leg')
  ap.add_argument('--pce-mask', choices=['both', '0', '1', 'nan'], default='both',
                  help='public_chunk_ens: run without / with the mask coordinate only (profiling: one kind of launch per traced process)')
  ap.add_argument('--config5-inits', type=int, default=366)
  ap.add_argument('--cpu-workers', type=int, default=0, help='worker processes of the multi-core CPU baseline (0 = os.cpu_count())')
  ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                  help='torch.distributed backend for N > 1; gloo combines through the host and lets several ranks share '
                       'one device (a plumbing check of the N > 1 path on a 1-GPU box, not a measurement)')
  ap.add_argument('--legs', default='all', help='comma list of main,configs1,ensemble,public_chunk,public_chunk_ens,spectrum,lat_fastest,config5,cpu '
                  '(profiling passes: one kernel shape per trace); default all')
  ap.add_argument('--small', action='store_true', help='tiny sizes (debugging only; not a valid measurement)')
  return ap.parse_args()


def self_launch(args):
  """`python bench.py --gpus N` from a bare shell: re-run under torch.distributed.run, one rank per GPU."""
  from weatherbenchx_amd import _hip
  ndev = _hip.device_count() if os.path.exists(_hip.lib_path()) else 0
  if ndev < args.gpus and not (args.backend == 'gloo' and ndev >= 1):
    print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'evals/s', 'n_gpus': args.gpus, 'steps': args.steps,
                      'warmup': args.warmup, 'skipped': True, 'higher_is_better': True,
                      'reason': f'--gpus {args.gpus} needs {args.gpus} visible devices, this box has {ndev} '
                                '(one rank per GPU: RCCL refuses two ranks on one device)'}))
    return 0
  import socket
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  return subprocess.call(cmd, env=env)


COLLECTIVE_CABI = 'wbx_acc_allreduce (C ABI, RCCL)'


def pick_collective(world, backend, environ):
  """Who moves the sums between the ranks: the library's own communicator on RCCL ranks unless WBX_COLLECTIVE=torch."""
  if world <= 1:
    return None
  if backend == 'nccl' and environ.get('WBX_COLLECTIVE', 'cabi') != 'torch':
    return COLLECTIVE_CABI
  return 'torch.distributed ' + backend


class Env:
  """Everything the legs share: process group, device, grid, generators."""

  def __init__(self, args):
    import torch
    import torch.distributed as dist
    from weatherbenchx_amd import _hip
    self.args, self.torch, self.dist = args, torch, dist
    self.world = int(os.environ.get('WORLD_SIZE', '1'))
    self.rank = int(os.environ.get('RANK', '0'))
    self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    self.device_index = self.local_rank % max(torch.cuda.device_count(), 1)  # (gloo plumbing runs may share a device)
    torch.cuda.set_device(self.device_index)
    self.dev = torch.device('cuda', self.device_index)
    if self.world > 1:
      if args.backend == 'nccl':
        dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.dev)
      else:
        dist.init_process_group('gloo', rank=self.rank, world_size=self.world)
    self.ctx = _hip.default_context(self.device_index)
    # The payload collective of every leg goes through the LIBRARY's RCCL communicator (wbx_comm_create / wbx_acc_allreduce: what
    # INTEGRATION.md documents for a host without torch); torch.distributed only carries the 128-byte unique id and the slot
    # tables (KBs, once).  WBX_COLLECTIVE=torch: dist.all_reduce on the same device buffer instead (the A/B).
    self.comm = None
    self.collective = pick_collective(self.world, args.backend, os.environ)
    if self.collective == COLLECTIVE_CABI:
      # (a failure to bind RCCL -- no librccl.so.1 to dlopen, a symbol missing -- is the same on every rank of a node; the ranks
      #  still agree on the outcome through the torch group before anybody takes a path the others do not)
      from weatherbenchx_amd import distributed
      why = None
      try:
        self.comm = distributed.CabiCommunicator.from_torch_group()
      except Exception as e:  # pylint: disable=broad-except
        why = f'{type(e).__name__}: {e}'
      ok = torch.tensor([0 if why else 1], device=self.dev, dtype=torch.int32)
      dist.all_reduce(ok, op=dist.ReduceOp.MIN)
      if int(ok.item()) == 0:
        if self.comm is not None:
          self.comm.close()
        self.comm = None
        self.collective = 'torch.distributed ' + args.backend + f' (the library\'s communicator could not be created on every rank: {why or "another rank"})'
    self.nlat, self.nlon = (721, 1440) if not args.small else (73, 144)
    self.lat = np.linspace(-90, 90, self.nlat)
    self.lon = np.linspace(0, 360, self.nlon, endpoint=False)
    self.set_layout(args.layout)
    self.gen = torch.Generator(device=self.dev)
    self.gen.manual_seed(1234 + self.rank)

  def set_layout(self, layout):
    self.layout = layout
    self.sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')

  def randn(self, shp, offset=0.0, scale=1.0, gen=None):
    t = self.torch.randn(shp, generator=gen or self.gen, device=self.dev, dtype=self.torch.float32)
    if scale != 1.0:
      t *= scale
    if offset != 0.0:
      t += offset
    return t

  def sp_shape(self):
    return tuple({'latitude': self.nlat, 'longitude': self.nlon}[d] for d in self.sp)

  def sync(self):
    self.torch.cuda.synchronize()
    self.ctx.synchronize()
    if self.world > 1:
      self.dist.barrier()
      self.torch.cuda.synchronize()

  def max_over_ranks(self, seconds):
    if self.world == 1:
      return seconds
    t = self.torch.tensor([seconds], device=self.dev if self.args.backend == 'nccl' else 'cpu', dtype=self.torch.float64)
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
    return float(t.item())


def _collective_record(env):
  """What the library's communicator says about the payload collectives so far: count, the last one's bytes and its own time
  on the library's stream (event pair around wbx_acc_allreduce), the mean."""
  if env.comm is None:
    return None
  t = env.comm.timings
  return {'rccl_ranks': env.comm.nranks, 'collectives': t['collectives'], 'bytes': t['bytes_last'],
          'us_last': None if t['us_last'] is None else round(t['us_last'], 1),
          'us_mean': round(t['us_total'] / t['collectives'], 1) if t['collectives'] else None}


def fresh(d):
  """New DataArray objects on the same payloads: nothing (statistics, plan results) is cached across steps."""
  from weatherbenchx_amd import xarray_lite as xr
  return {k: xr.DataArray(v.data, dims=v.dims, coords={c: v[c].values for c in v.dims}) for k, v in d.items()}


def pipelined(launch, finish, n):
  """n steps, one deep: step k+1 is launched before step k is finished (host bookkeeping overlaps the kernels)."""
  out, pending = None, None
  for _ in range(n):
    cur = launch()
    if pending is not None:
      out = finish(pending)
    pending = cur
  if pending is not None:
    out = finish(pending)
  return out


def kernel_roofline(name, ms, nbytes, traffic=None):
  ach = nbytes / (ms * 1e-3) / 1e9
  return {'bound': 'hbm', 'kernel': name, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
          'frac': round(ach / HBM_PEAK_GBS, 4), 'kernel_ms': round(ms, 4), 'algorithmic_bytes_per_launch': int(nbytes),
          'traffic': traffic}


def pmc_traffic(kernel_key, enabled=True, variant=None):
  """HBM traffic per launch from the separate rocprofv3 --pmc passes of this same command (FETCH_SIZE x2 on gfx950,
  + WRITE_SIZE), committed under profiles/ -- PMC collection cannot run inside the timed process."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
  if not files or not enabled:
    return None
  table = json.load(open(files[-1]))
  entry = table.get(kernel_key.split(' (')[0] + (f'@{variant}' if variant else ''))
  if not entry:
    return None
  # a figure measured on ANOTHER build of the library says nothing about this one (round 4: every entry carries the md5 of the
  # .so it was collected with; files of earlier rounds carry none and are not replayed against a newer library)
  stamp = entry.get('library_md5') or table.get('_library_md5')
  if stamp != _library_md5():
    return None
  return entry.get('traffic_bytes_per_launch')


_LIB_MD5 = []


def _library_md5():
  if not _LIB_MD5:
    import hashlib
    from weatherbenchx_amd import _hip
    path = _hip.lib_path()
    _LIB_MD5.append(hashlib.md5(open(path, 'rb').read()).hexdigest() if os.path.exists(path) else None)
  return _LIB_MD5[0]


# ---- main line: the north_star field ---------------------------------------------------------------------------------------
def ens_kernel_name(e, m=51):
  """The kernel wbx_ens_partial launches for this event (the dispatch rule of csrc/wbx_ens_impl.hpp restated)."""
  if e.get('flags', 0) & 8:  # WBX_FLAG_SKIPNA_ENS: per-point member counts
    if os.environ.get('WBX_ENS_SKIPNA_GENERIC', '0') != '0' or m > 64:
      return f"s1_{'xk' if e.get('x_kept') else 'xr'}_kernel<EnsOpGeneric<float>,1> (pair form over the valid members re-read from memory, fp64)"
    if (os.environ.get('WBX_ENS_PIPE', '1') != '0' and os.environ.get('WBX_ENS_PIPE_SKIPNA', '1') != '0' and e.get('block') == 64
        and not e.get('x_kept') and not (e.get('flags', 0) & 3) and not e.get('x_weighted')):
      return (f'ens_pipe_kernel<{m},true,SKIPNA_SORT> (r5: NaN members -> +inf, sorted in registers, rank form over the first n members '
              'with per-point n, fp64 sums; next tile through LDS-DMA, 3 waves per SIMD)')
    return (f"s1_{'xk' if e.get('x_kept') else 'xr'}_kernel<EnsOpF32<{m},true,SKIPNA_SORT>,1> (NaN members -> +inf, sorted in registers, rank form "
            'over the first n members with per-point n, fp64 sums)')
  if e.get('algo') == 1:
    return f's1_xr_kernel<EnsOpF32<{m},true,PAIRWISE>,1> (register-tiled O(M^2) pair form)'
  piped = (os.environ.get('WBX_ENS_PIPE', '1') != '0' and e.get('block') == 64 and not e.get('x_kept') and not (e.get('flags', 0) & 11)
           and (e.get('flat') or not e.get('x_weighted')))
  if piped and e.get('flat'):
    return (f'ens_pipe_kernel<{m},true,SORT,FLAT> (rank form, contiguous planes with folded latitude weights, next tile through '
            'LDS-DMA, fp32 chain sums)')
  if piped:
    return f'ens_pipe_kernel<{m},true,SORT> (rank form, next tile through LDS-DMA, fp32 chain sums)'
  if e.get('flat'):
    return f's1_xf1_kernel<EnsOpF32<{m},true,SORT>> (rank form, folded latitude weights, flat sweep, one point per lane)'
  return f"s1_{'xk' if e.get('x_kept') else 'xr'}_kernel<EnsOpF32<{m},true,SORT>,1>"


def oracle_level_check(env, ens_level, t_level, got):
  """One [51, spatial] level of the timed field on the host through the oracle (float64, reference structure): the four
  metrics of the suite with GridAreaWeighting over (latitude, longitude) -> max relative error of `got` (asserted < 1e-6)."""
  from oracle import wbx_oracle as O
  pv, tvh = ens_level.cpu().numpy(), t_level.cpu().numpy()
  pd, td = ('number',) + env.sp, env.sp
  w = [(O.grid_area_weights(env.lat), ('latitude',))]
  mean = lambda lane: (lambda a: float(a[0] / a[1]))(O.aggregate(lane[0], lane[1], ['latitude', 'longitude'], weights=w))
  skill, spread = mean(O.crps_skill(pv, pd, tvh, td, 'number')), mean(O.crps_spread(pv, pd, 'number', fair=True, use_sort=True))
  var, uemse = mean(O.ensemble_variance(pv, pd, 'number')), mean(O.unbiased_ensemble_mean_squared_error(pv, pd, tvh, td, 'number'))
  emse = mean(O.ensemble_mean_squared_error(pv, pd, tvh, td, 'number'))
  want = {'crps': O.crps(skill, spread), 'unbiased_spread_skill': float(np.sqrt(var / uemse)), 'unbiased_mean_rmse': float(np.sqrt(uemse)),
          'mean_rmse': float(np.sqrt(emse))}
  err = {k: abs(float(got[k]) / want[k] - 1.0) for k in want}
  assert max(err.values()) < 1e-6, (err, got, want)
  return {'oracle_max_rel_err': max(err.values()), 'oracle_rel_err': err}


def main_leg(env):
  """ONE f32[1 init, 37 level, 51 member, lat, lon] forecast per rank against f32[1, 37, lat, lon] targets.

  One launch per step, every step on the SAME stream (`--main-streams 1`): a step's kernel then runs alone on the chip, so
  the duration between its two timing marks is the launch's own and is what rocprofv3 reports for it.  (Dealing steps to
  two streams, as the chunk loops do for ensemble kernels, lets the tail of one launch overlap the head of the next --
  about 3 % per step here -- but two overlapping launches each read as ~2 ms in any per-kernel timing.)"""
  from weatherbenchx_amd import engine
  saved = engine.ALTERNATE_STREAMS
  engine.ALTERNATE_STREAMS = env.args.main_streams == 2
  try:
    return _main_leg(env)
  finally:
    engine.ALTERNATE_STREAMS = saved


def _main_leg(env):
  from weatherbenchx_amd import aggregation, distributed, engine, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  args, m = env.args, 51
  nlev = 37 if not args.small else 3
  sp_shape = env.sp_shape()
  # every rank's field is another (init, lead) time slice: its own init_time label, the sums combine over init_time
  init_time = np.array([np.datetime64('2020-01-01T00', 'ns') + env.rank * np.timedelta64(24, 'h')])
  cs = {'init_time': init_time, 'level': np.arange(nlev), 'latitude': env.lat, 'longitude': env.lon}
  # exchangeable synthetic ensemble: members and the target are independent N(0, 1) draws around a common state, so the
  # unbiased spread/skill ratio is ~1 (a target that IS the ensemble centre makes the unbiased MSE vanish)
  tv = env.randn((1, nlev) + sp_shape, 280.0)
  ens = env.randn((1, nlev, m) + sp_shape)
  ens += tv[:, :, None]
  tv += env.randn((1, nlev) + sp_shape)
  pe = {'t': xr.DataArray(ens, dims=('init_time', 'level', 'number') + env.sp, coords=cs)}
  te = {'t': xr.DataArray(tv, dims=('init_time', 'level') + env.sp, coords=cs)}
  env.torch.cuda.synchronize()
  metrics = ensemble_suite()
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  plan_box = [None]

  def stats():
    return metrics_base.compute_unique_statistics_for_all_metrics(metrics, fresh(pe), fresh(te))

  if env.world == 1:
    def launch():  # Statistic.compute -> Aggregator.aggregate_statistics: enqueues the kernels and the read-back
      return agg.aggregate_statistics(stats())

    def finish(state):  # waits for THAT step's sums only, then the metric values on the host
      return state.metric_values(metrics)

    def run(n):
      with engine.deferred_results():
        return pipelined(launch, finish, n)
  else:
    def launch():  # the step's sums stay in HBM ...
      acc = engine.Accumulation()
      with engine.accumulate_results(acc):
        return agg.aggregate_statistics(stats()), acc

    def finish(pair):  # ... and are summed over the ranks with ONE all-reduce of the device buffer, read back once
      state, plan_box[0] = distributed.resolve_state(pair[0], pair[1], plan=plan_box[0], comm=env.comm)
      return state.metric_values(metrics)

    def run(n):
      return pipelined(launch, finish, n)

  # steady state: a step is ~1.3 ms, so W = 5 warm-up steps are 7 ms of load -- the device is still
  # climbing to its sustained clocks then (measured: 1.46 / 1.34 / 1.23 ms per step with 10+3 / 20+5 / 50+20 steps on one
  # box, the kernel itself 1.25 ms throughout).  Untimed steps of the same loop run for `--prewarm-ms` first; then the W
  # warm-up steps and EXACTLY K timed steps as the contract says.
  run(1)  # (the first call builds and uploads the launch plan)
  env.sync()
  tp = time.perf_counter()
  run(8)
  env.sync()
  per_step = env.max_over_ranks((time.perf_counter() - tp) / 8)  # the same count on every rank: each step holds a collective
  prewarm_steps = 8 + max(0, int(np.ceil(args.prewarm_ms * 1e-3 / per_step)) - 8) if args.prewarm_ms > 0 else 8
  run(prewarm_steps - 8)
  out = run(args.warmup)
  env.sync()
  c0 = plan_box[0].collectives if plan_box[0] is not None else 0
  # every launch of the timed region sits between two timing marks on its launch stream (wbx_mark: two event records per
  # launch, nothing waits); they are read after the region's closing synchronisation
  engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = [], True
  t0 = time.perf_counter()
  out = run(args.steps)
  env.sync()
  dt = env.max_over_ranks(time.perf_counter() - t0)
  log = engine.resolve_event_marks([e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens'])
  engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = None, False
  ms_per_step = dt / args.steps * 1e3
  points = nlev * env.nlat * env.nlon
  value = points * len(metrics) * env.world / (ms_per_step * 1e-3)

  ms_list = sorted(e['ms'] for e in log)
  k_ms = float(np.mean(ms_list))  # the average launch duration over the K timed steps (rank 0)
  kname = ens_kernel_name(log[0], m)
  roofline = kernel_roofline(kname, k_ms, points * (m + 1) * 4, pmc_traffic(kname.split('<')[0], nlev == 37 and not args.small, f'main@{env.layout}'))
  roofline['kernel_ms_source'] = (f'HIP events around each of the {len(log)} launches of the timed region on their launch stream '
                                  '(wbx_mark), mean; the rocprofv3 --kernel-trace median of the same command is in profiles/')
  roofline['kernel_ms_median'] = round(float(np.median(ms_list)), 4)
  roofline['kernel_ms_min_max'] = [round(ms_list[0], 4), round(ms_list[-1], 4)]
  roofline['launches_per_step'] = len(log) // max(args.steps, 1)
  roofline['bytes_per_point'] = (m + 1) * 4
  roofline['traffic_source'] = 'profiles/r*_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; replayed, not measured in this run)'

  result = {
      'metric': METRIC, 'value': value, 'unit': 'evals/s', 'n_gpus': env.world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32+f64',
      'data': 'synthetic',
      'config': {'workload': f'north_star: area-weighted CRPS(fair) + unbiased spread/skill + unbiased-mean RMSE + ensemble-mean RMSE on ONE '
                             f'f32[1 init,{nlev} level,{m} member,{env.nlat},{env.nlon}] forecast per GPU vs f32[1,{nlev},{env.nlat},{env.nlon}] '
                             f'targets, reduce (init_time,latitude,longitude), GridAreaWeighting, {env.layout}',
                 'points_per_step_per_gpu': points, 'members': m, 'metrics': list(metrics), 'input_dtype': 'f32',
                 'arithmetic': 'sorting network + fp32 chain sums per point (worst case 9 x 2^-24 relative, wbx_ens_impl.hpp), '
                               'fp64 sums across points',
                 'accumulators': 'f64', 'layout': env.layout,
                 'prewarm': f'{prewarm_steps} untimed steps ({args.prewarm_ms:.0f} ms under load) before the {args.warmup} warm-up steps',
                 'streams': args.main_streams,
                 'host_pipeline': 'steps overlapped one deep; ' + ('deferred read-back (engine.deferred_results)' if env.world == 1 else
                                  'sums accumulated in HBM (engine.Accumulation), all-reduced on the device buffer'),
                 'sharding': f'{env.world} x one (init, lead) field per rank, 1 all-reduce/step',
                 'rccl_ranks': env.world if (env.world > 1 and args.backend == 'nccl') else 0, 'backend': args.backend if env.world > 1 else None,
                 'collectives_per_step': ((plan_box[0].collectives - c0) / args.steps) if plan_box[0] is not None else 0,
                 'collective_backend': env.collective, 'collective': _collective_record(env)},
      'roofline': roofline,
  }
  # outside the timed region: one sampled level of the timed field against the float64 oracle, every metric of the suite
  # (the kernel's per-level sums are independent: a level checks the kernel, the planes' addressing and the weights)
  crps = float(np.asarray(out['crps.t'].values).mean())
  ssr = float(np.asarray(out['unbiased_spread_skill.t'].values).mean())
  result['check'] = {'crps_mean': crps, 'unbiased_spread_skill_mean': ssr, 'mean_rmse': float(np.asarray(out['mean_rmse.t'].values).mean())}
  if env.world == 1:
    result['check'].update(oracle_level_check(env, ens[0, nlev // 2], tv[0, nlev // 2],
                                              {k: np.asarray(out[f'{k}.t'].values).reshape(-1)[nlev // 2] for k in metrics}), level=nlev // 2)
  else:  # (N > 1: the sums combine every rank's field; an exchangeable N(0, 1) ensemble has CRPS ~ 0.5642, spread/skill ~ 1)
    assert np.isfinite(crps) and abs(crps - 0.5642) < 0.01 and abs(ssr - 1.0) < 0.01, (crps, ssr)
  del pe, te, ens, tv
  return result


# ---- configs[1]: the deterministic suite (main line of rounds 1-2), N = 1 ----------------------------------------------------
def configs1_leg(env):
  from weatherbenchx_amd import aggregation, engine, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  from weatherbenchx_amd.metrics import deterministic
  args = env.args
  ni, nl, nlev = (args.inits, args.leads, args.levels) if not args.small else (4, 3, 2)
  init_time = np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(24, 'h')
  lead_time = (np.arange(nl) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  ndoy = int(ni + (nl * 6) // 24 + 2)
  coords = {'init_time': init_time, 'lead_time': lead_time, 'level': np.arange(nlev), 'latitude': env.lat, 'longitude': env.lon}
  dims = ('init_time', 'lead_time', 'level') + env.sp
  cdims = ('dayofyear', 'hour', 'level') + env.sp
  shape = tuple(len(coords[d]) for d in dims)
  clim_t = env.randn((ndoy, 4) + shape[2:], 280.0, 10.0)
  clim = xr.Dataset({'z': xr.DataArray(clim_t, dims=cdims, coords={
      'dayofyear': np.arange(1, ndoy + 1), 'hour': np.array([0, 6, 12, 18]), **{d: coords[d] for d in cdims[2:]}})})
  p_t, t_t = env.randn(shape, 280.0), env.randn(shape, 280.0)
  p = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
  t = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
  env.torch.cuda.synchronize()
  metrics = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'mae': deterministic.MAE(),
             'bias': deterministic.Bias(), 'acc': deterministic.ACC(clim),
             'prediction_activity': deterministic.PredictionActivity(clim)}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  def stats():
    return metrics_base.compute_unique_statistics_for_all_metrics(metrics, fresh(p), fresh(t))

  def launch():  # Statistic.compute -> Aggregator.aggregate_statistics: enqueues the kernels and the read-back
    return agg.aggregate_statistics(stats())

  def finish(state):  # waits for THAT step's sums only, then the metric values on the host
    return state.metric_values(metrics)

  def run(n):
    with engine.deferred_results():
      return pipelined(launch, finish, n)

  out = run(args.warmup)
  env.sync()
  t0 = time.perf_counter()
  out = run(args.steps)
  env.sync()
  dt = time.perf_counter() - t0
  ms_per_step = dt / args.steps * 1e3
  points = int(np.prod(shape, dtype=np.int64))
  value = points * len(metrics) / (ms_per_step * 1e-3)

  # roofline leg: HIP events around the dominant (stage-1) kernel, separate pass, 10 launches per event pair
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 10
  for _ in range(3):
    agg.aggregate_statistics(stats()).metric_values(metrics)
  log = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'det']
  engine.S1_EVENT_LOG = None
  k_ms = float(np.mean([e['ms'] for e in log]))
  if log[0].get('flat'):
    kname = 's1_xf_kernel<DetOp<float,DET6>> (latitude weights folded into stage 1, flat float4 sweep)'
  elif log[0].get('x_weighted'):
    kname = 's1_xr_kernel<XWeighted<DetOp<float,DET6>>,1> (latitude weights folded into stage 1)'
  elif log[0].get('plane_rows'):
    kname = f"s1_xp_kernel<DetOp<float,DET6>> (LDS plane mode, R={log[0]['plane_rows']})"
  elif log[0]['x_kept']:
    kname = f"s1_xk_kernel<DetOp<float,DET6>,{log[0]['vec']}>"
  else:
    kname = f"s1_xr_kernel<DetOp<float,DET6>,{log[0]['vec']}>"
  full = (not args.small) and (ni, nl, nlev) == (40, 10, 5)
  roofline = kernel_roofline(kname, k_ms, points * 12, pmc_traffic(kname, full, f'configs1@{env.layout}'))
  roofline['traffic_source'] = 'profiles/r*_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'

  result = {
      'workload': f'configs[1]: f32[{ni} init,{nl} lead,{nlev} level,{env.nlat},{env.nlon}] p,t + (doy,hour) climatology, '
                  f'rmse/mse/mae/bias/acc/activity, reduce (init_time,latitude,longitude), GridAreaWeighting, {env.layout}',
      'value': value, 'unit': 'evals/s', 'ms_per_step': ms_per_step, 'points_per_step': points, 'metrics': list(metrics),
      'roofline': roofline,
  }
  # sanity: values must be finite and physically plausible (sigma=1 errors -> rmse ~ sqrt(2))
  r = float(np.asarray(out['rmse.z'].values).mean())
  assert np.isfinite(r) and abs(r - np.sqrt(2.0)) < 0.01, r
  result['check'] = {'rmse_mean': r, 'acc_mean': float(np.asarray(out['acc.z'].values).mean())}
  keep = (p_t, t_t, clim_t, coords, nlev, ni, nl, len(metrics))
  return result, keep


# ---- ensemble legs ---------------------------------------------------------------------------------------------------
def ensemble_suite():
  from weatherbenchx_amd.metrics import deterministic, probabilistic, wrappers
  return {'crps': probabilistic.CRPSEnsemble(use_sort=True),
          'unbiased_spread_skill': probabilistic.UnbiasedSpreadSkillRatio(),
          'unbiased_mean_rmse': probabilistic.UnbiasedEnsembleMeanRMSE(),
          'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')])}


def ens_leg(env, lead_dim, nlead, nvar, name, describe):
  """`nvar` variables x f32[nlead, 51, lat, lon] against f32[nlead, lat, lon]: one fused launch per variable."""
  from weatherbenchx_amd import aggregation, engine, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  args, m = env.args, 51
  sp_shape = env.sp_shape()
  cs = {'latitude': env.lat, 'longitude': env.lon}
  # exchangeable synthetic ensemble: members and the target are independent N(0, 1) draws around a common state, so the
  # unbiased spread/skill ratio is ~1 (a target that IS the ensemble centre makes the unbiased MSE vanish)
  tv = {f'v{i}': env.randn((nlead,) + sp_shape, 280.0) for i in range(nvar)}
  pe, te = {}, {}
  for k, v in tv.items():
    ens = env.randn((nlead, m) + sp_shape)
    ens += v[:, None]
    v += env.randn((nlead,) + sp_shape)
    pe[k] = xr.DataArray(ens, dims=(lead_dim, 'number') + env.sp, coords=cs)
    te[k] = xr.DataArray(v, dims=(lead_dim,) + env.sp, coords=cs)
  env.torch.cuda.synchronize()
  emetrics = ensemble_suite()
  eagg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])

  def launch():
    return eagg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(emetrics, fresh(pe), fresh(te)))

  def run(n):
    with engine.deferred_results():
      return pipelined(launch, lambda s: s.metric_values(emetrics), n)
  run(2)
  env.sync()
  t0 = time.perf_counter()
  eout = run(args.steps)
  env.sync()
  e_ms = (time.perf_counter() - t0) / args.steps * 1e3
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 20 if nlead <= 8 else 5
  for _ in range(3):
    launch().metric_values(emetrics)
  elog = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens']
  # (a) the reference-DEFAULT CRPSEnsemble() (use_sort=False, probabilistic.py:644) through the API: launches per variable
  # and their cost -- it shares the rank-form launch since round 3 (lazy.ens_statistic); (b) the north_star's pairwise
  # |x_i - x_j| form as a kernel: lazy.PAIR_FORM_KERNEL routes use_sort=False to the register-tiled O(M^2) pair kernel, and
  # only events of THAT algorithm are counted (round 2 averaged rank- and pair-form launches here).
  from weatherbenchx_amd import lazy as _lazy
  from weatherbenchx_amd.metrics import probabilistic as _prob
  pmet = {'crps_default': _prob.CRPSEnsemble()}
  k0 = next(iter(pe))

  def default_crps():
    return eagg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(
        pmet, fresh({k0: pe[k0]}), fresh({k0: te[k0]}))).metric_values(pmet)
  engine.S1_EVENT_LOG = []
  for _ in range(2):
    dout = default_crps()
  dlog = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens']
  engine.S1_EVENT_LOG = []
  was = _lazy.PAIR_FORM_KERNEL
  _lazy.PAIR_FORM_KERNEL = True
  try:
    for _ in range(2):
      pout = default_crps()
  finally:
    _lazy.PAIR_FORM_KERNEL = was
  plog = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens' and e.get('algo') == 1]
  # (c) skipna_ensemble=True (probabilistic.py:133-145: NaN members are left out point by point) only exists in pair form with
  # per-point member counts: the generic from-memory kernel (WBX_FLAG_SKIPNA_ENS), timed here on the same variable
  smet = {'crps_skipna': _prob.CRPSEnsemble(skipna_ensemble=True)}
  engine.S1_EVENT_LOG = []
  for _ in range(2):
    sout = eagg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(
        smet, fresh({k0: pe[k0]}), fresh({k0: te[k0]}))).metric_values(smet)
  slog = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens']
  engine.S1_EVENT_LOG = None
  epoints = nlead * env.nlat * env.nlon  # per variable launch
  ek_ms = float(np.median([e['ms'] for e in elog]))
  kname = ens_kernel_name(elog[0], m)
  out = {'workload': describe, 'value': epoints * nvar * len(emetrics) / (e_ms * 1e-3), 'unit': 'evals/s', 'ms_per_step': e_ms,
         'metrics': list(emetrics),
         'roofline': kernel_roofline(kname, ek_ms, epoints * (m + 1) * 4, pmc_traffic(kname.split('<')[0], nlead == 8 and not args.small, f'ensemble@{env.layout}')),
         'default_crps_ensemble': {'what': 'CRPSEnsemble() with the reference defaults (use_sort=False) on one variable through the API',
                                   'ensemble_launches_per_variable': len(dlog) // 2,
                                   'kernel_ms_per_variable': round(float(np.sum([e['ms'] for e in dlog]) / 2), 4),
                                   'frac_of_hbm_peak': round(epoints * (m + 1) * 4 / (float(np.sum([e['ms'] for e in dlog]) / 2) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   'crps': float(np.asarray(dout[f'crps_default.{k0}'].values).mean())},
         'pairwise_form': (dict(kernel_roofline(ens_kernel_name(plog[0], m) + ' via lazy.PAIR_FORM_KERNEL', float(np.median([e['ms'] for e in plog])),
                                                epoints * (m + 1) * 4,
                                                pmc_traffic(ens_kernel_name(plog[0], m), nlead == 8 and not args.small, f'ensemble@{env.layout}')),
                                crps=float(np.asarray(pout[f'crps_default.{k0}'].values).mean())) if plog else None),
         'skipna_ensemble': ({'what': 'CRPSEnsemble(skipna_ensemble=True) on one variable through the API (WBX_FLAG_SKIPNA_ENS: per-point member '
                                      'counts; round 4: register-resident rank form over the valid members -- the generic from-memory pair '
                                      'form it replaces took 10.7 ms = 2 % of the HBM peak; round 5: on the pipelined sweep 0.455 -> 0.424 ms; members past the n-th = the shift, compile-time rank coefficients, v_rcp_f64 + Newton for 1/n: 209 -> 151 VGPRs, three waves per SIMD, 0.378 ms; missing members = copies of the shift in the sorted order (rank correction k (sum |e| + sum e)), no selects or class tests: 0.336 ms)',
                              'launches_per_variable': len(slog) // 2,
                              'launch': {k: slog[0].get(k) for k in ('block', 'grid', 'flags', 'x_kept', 'x_weighted', 'algo')},
                              'roofline': kernel_roofline(ens_kernel_name(slog[0], m), float(np.sum([e['ms'] for e in slog]) / 2),
                                                          epoints * (m + 1) * 4,
                                                          pmc_traffic('ens_pipe_kernel_skipna' if ens_kernel_name(slog[0], m).startswith('ens_pipe')
                                                                      else ens_kernel_name(slog[0], m),
                                                                      nlead == 8 and not args.small, f'ensemble@{env.layout}')),
                              'crps': float(np.asarray(sout[f'crps_skipna.{k0}'].values).mean())} if slog else None),
         'check': {'crps_v0_mean': float(np.asarray(eout['crps.v0'].values).mean()),
                   'spread_skill_v0_mean': float(np.asarray(eout['unbiased_spread_skill.v0'].values).mean())}}
  del pe, te, tv
  return name, out


def chunk_loop_ms(env, make_chunk, metrics, agg, lead_time, nchunks):
  """The chunk loop of the pipeline (pipeline.evaluate_chunks: the body of beam_pipeline.py:161-250 per chunk, sums kept in HBM,
  metric values once at the end) over `nchunks` one-init chunks of resident data: -> (ms per chunk, chunk-record statistics,
  final metric values).  `make_chunk(i, init_times)` -> (predictions, targets).  Steady state = chunk records (replay.py):
  the first chunk builds plans / tables / accumulator slots, the second of each kind is recorded, the rest are ONE library call."""
  from weatherbenchx_amd import pipeline, replay, time_chunks
  inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(nchunks) * np.timedelta64(24, 'h')
  index = {int(t.astype('int64')): i for i, t in enumerate(inits)}

  def load(ic, lc):
    return make_chunk(index[int(ic[0].astype('int64'))], ic)

  def run(k):
    out = pipeline.evaluate_chunks(time_chunks.TimeChunks(inits[:k], lead_time, init_time_chunk_size=1), load, metrics, agg)
    return out[None].metric_values(metrics)
  run(8)
  env.sync()
  best, vals, stats = None, None, None
  for _ in range(2):  # two jobs, the faster one (the first may still meet a new time label's gather table)
    replay.reset_stats()
    t0 = time.perf_counter()
    vals = run(nchunks)
    env.sync()
    ms = (time.perf_counter() - t0) / nchunks * 1e3
    if best is None or ms < best:
      best, stats = ms, {k: v for k, v in replay.STATS.items() if k != 'refusals'}
      stats['refusals'] = list(replay.STATS['refusals'][:2])
  return best, stats, vals


# ---- public-benchmark chunk ----------------------------------------------------------------------------------------
def public_chunk_leg(env):
  from weatherbenchx_amd import aggregation, binning, engine, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  from weatherbenchx_amd.metrics import deterministic
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  from wb_regions import REGIONS  # the reference's region table restated as data
  args = env.args
  pl, plev = (12, 13) if not args.small else (3, 2)
  pdims = ('init_time', 'lead_time', 'level') + env.sp
  cdims = ('dayofyear', 'hour', 'level') + env.sp
  pcoords = {'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
             'lead_time': (np.arange(pl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
             'level': np.arange(plev), 'latitude': env.lat, 'longitude': env.lon}
  pshape = tuple(len(pcoords[d]) for d in pdims)
  pp_t, pt_t = env.randn(pshape, 280.0), env.randn(pshape, 280.0)
  pclim = xr.Dataset({'z': xr.DataArray(env.randn((10, 4) + pshape[2:], 280.0, 10.0), dims=cdims, coords={
      'dayofyear': np.arange(1, 11), 'hour': np.array([0, 6, 12, 18]), **{d: pcoords[d] for d in cdims[2:]}})})
  land = (np.sin(np.deg2rad(env.lon) * 3)[None, :] * np.cos(np.deg2rad(env.lat) * 2.5)[:, None]) > 0.35
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': env.lat, 'longitude': env.lon})
  pmetrics = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'bias': deterministic.Bias(),
              'acc': deterministic.ACC(pclim), 'prediction_activity': deterministic.PredictionActivity(pclim)}
  pagg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                                bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
  env.torch.cuda.synchronize()

  def plaunch():
    return pagg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(
        pmetrics, {'z': xr.DataArray(pp_t, dims=pdims, coords=pcoords)}, {'z': xr.DataArray(pt_t, dims=pdims, coords=pcoords)}))

  def prun(n):
    with engine.deferred_results():
      return pipelined(plaunch, lambda s: s.metric_values(pmetrics), n)
  prun(args.warmup)
  env.sync()
  t0 = time.perf_counter()
  pout = prun(args.steps * 2)
  env.sync()
  p_ms = (time.perf_counter() - t0) / (args.steps * 2) * 1e3
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 5
  plaunch().wait()
  plog = list(engine.S1_EVENT_LOG)
  engine.S1_EVENT_LOG = None
  ppoints = int(np.prod(pshape))
  k_ms = float(np.sum([e['ms'] for e in plog]))
  # the chunk loop (sums stay in HBM, values at the end): what a job over many such chunks pays per chunk
  ring = np.datetime64('2020-01-01T00', 'ns') + np.arange(4) * np.timedelta64(24, 'h')  # (the climatology holds ten days)

  def make_chunk(i, ic):
    cs = dict(pcoords, init_time=ic, valid_time=(('init_time', 'lead_time'), ring[i % 4] + pcoords['lead_time'][None, :]))
    return {'z': xr.DataArray(pp_t, dims=pdims, coords=cs)}, {'z': xr.DataArray(pt_t, dims=pdims, coords=cs)}
  loop_ms, loop_stats, _ = chunk_loop_ms(env, make_chunk, pmetrics, pagg, pcoords['lead_time'], int(os.environ.get('WBX_BENCH_CHUNKS', 100 if not args.small else 12)))
  return {'workload': f'public benchmark chunk: f32[1 init,{pl} lead,{plev} level,{env.nlat},{env.nlon}] p,t + climatology, '
                      f'rmse/mse/bias/acc/activity, GridAreaWeighting, {len(REGIONS)} regions x land/sea = '
                      f'{2 * len(REGIONS)} bins, masked=True, {env.layout}',
          'ms_per_chunk': loop_ms, 'value': ppoints * len(pmetrics) / (loop_ms * 1e-3), 'unit': 'evals/s',
          'ms_per_chunk_is': 'pipeline.evaluate_chunks over 100 such chunks (sums in HBM, values at the end; chunk records on)',
          'chunk_over_kernel': round(loop_ms / k_ms, 3), 'chunk_records': loop_stats,
          'ms_per_chunk_values_every_chunk': p_ms,
          'kernels': [e.get('kind') for e in plog],
          'roofline': dict(kernel_roofline('wbx_det_binned (memset + det_atoms_kernel + slot kernel for overflow patches + finish)',
                                           k_ms, ppoints * 12,
                                           pmc_traffic(f"det_atoms_kernel<float,DET6,MM=0,PD=4,WM={2 if env.layout == 'lon_fastest' else 1}>",
                                                       not args.small, f'public_chunk@{env.layout}')),
                           traffic_note='PMC pass of the main kernel only (bench.py public_chunk leg / tools/kbench_binned.py)'),
          'check': {'acc_first': float(np.asarray(pout['acc.z'].values).reshape(-1)[0])}}


# ---- public-benchmark chunk, probabilistic configuration --------------------------------------------------------------
def public_chunk_ens_leg(env, ifs_layout=False, with_mask=True):
  """The reference's probabilistic evaluation (public_benchmark/run_benchmark_evaluation.py:341-354, 365-382): the ensemble
  suite under Regions(17) x land/sea = 34 bins, GridAreaWeighting, masked=True, on one f32[1 init, 8 lead, 51 member, lat, lon]
  chunk.  The whole suite is ONE wbx_ens_binned launch per variable -- asserted -- also when (`with_mask`) the targets carry a
  (latitude, longitude) `mask` coordinate: the reference then masks skill / unbiased MSE / mean MSE and leaves the statistics
  of the predictions alone (spread, variance: no mask coordinate, probabilistic.py:165-273) unmasked, and the kernel yields both
  sets from one pass (masked-out points are accumulated under their atom's twin).  The roofline is that of the launch.  `ifs_layout`: the recorded IFS-ENS dim order (init, number, lead, longitude, latitude),
  docs/source/how_to/metric_wrappers.ipynb:955-964."""
  from weatherbenchx_amd import aggregation, binning, engine, weighting
  from weatherbenchx_amd import data as wdata
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  from wb_regions import REGIONS  # the reference's region table restated as data
  args, m = env.args, 51
  nl = 8 if not args.small else 2
  sp_shape = env.sp_shape()
  coords = {'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
            'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'), 'latitude': env.lat, 'longitude': env.lon}
  if ifs_layout:
    pdims, tdims = ('init_time', 'number', 'lead_time') + env.sp, ('init_time', 'lead_time') + env.sp
    tv = env.randn((1, nl) + sp_shape, 280.0)
    ens = env.randn((1, m, nl) + sp_shape)
    ens += tv[:, None]
  else:
    pdims, tdims = ('init_time', 'lead_time', 'number') + env.sp, ('init_time', 'lead_time') + env.sp
    tv = env.randn((1, nl) + sp_shape, 280.0)
    ens = env.randn((1, nl, m) + sp_shape)
    ens += tv[:, :, None]
  tv += env.randn((1, nl) + sp_shape)
  land = (np.sin(np.deg2rad(env.lon) * 3)[None, :] * np.cos(np.deg2rad(env.lat) * 2.5)[:, None]
          + 0.3 * np.sin(np.deg2rad(env.lon) * 17)[None, :] * np.sin(np.deg2rad(env.lat) * 13)[:, None]) > 0.35
  valid = ~((np.abs(env.lat)[:, None] > 80) & (np.cos(np.deg2rad(env.lon) * 5)[None, :] > 0.2))  # a NaN-mask-like hole at the poles
  nan_mask = with_mask == 'nan'
  if nan_mask:
    # what the reference's loaders do (data_loaders/base.py:25-56): NaN targets + `mask = ~isnan(targets)` over EVERY dim of
    # the targets -- here a polar hole that grows with the lead time, so the mask has a lead_time stride
    holes = np.stack([(np.abs(env.lat)[:, None] > 80 - 2 * l) & (np.cos(np.deg2rad(env.lon) * (5 + l))[None, :] > 0.2) for l in range(nl)])
    hv = holes if env.sp == ('latitude', 'longitude') else np.ascontiguousarray(np.swapaxes(holes, 1, 2))
    tv[0][env.torch.as_tensor(hv, device=env.dev)] = float('nan')
    nan_valid = ~holes
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': env.lat, 'longitude': env.lon})
  mv = valid if env.sp == ('latitude', 'longitude') else np.ascontiguousarray(valid.T)
  mask_da = xr.DataArray(env.torch.as_tensor(mv, device=env.dev), dims=env.sp, coords={'latitude': env.lat, 'longitude': env.lon})
  nan_mask_da = None
  if nan_mask:  # the loader's step (add_nan_mask_to_data builds the mask in HBM with wbx_notnan_mask): once per chunk, not timed
    nan_mask_da = wdata.add_nan_mask_to_data({'v': xr.DataArray(tv, dims=tdims, coords={k: v for k, v in coords.items() if k in tdims})})['v'].coords['mask']
    assert tuple(nan_mask_da.dims) == tdims
  metrics = ensemble_suite()
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
  env.torch.cuda.synchronize()

  def launch():
    p = xr.DataArray(ens, dims=pdims, coords={k: v for k, v in coords.items() if k in pdims})
    t = xr.DataArray(tv, dims=tdims, coords={k: v for k, v in coords.items() if k in tdims})
    if nan_mask:
      t = t.assign_coords(mask=nan_mask_da)
    elif with_mask:
      t = t.assign_coords(mask=mask_da)
    return agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': p}, {'v': t}))

  def run(n):
    with engine.deferred_results():
      return pipelined(launch, lambda s: s.metric_values(metrics), n)
  run(max(args.warmup, 2))
  env.sync()
  t0 = time.perf_counter()
  out = run(args.steps * 2)
  env.sync()
  ms_chunk = (time.perf_counter() - t0) / (args.steps * 2) * 1e3
  # the launches of one chunk, each alone on its stream between HIP events (synchronising)
  saved = engine.ALTERNATE_STREAMS
  engine.ALTERNATE_STREAMS = False
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 10
  try:
    launch().wait()
    log = list(engine.S1_EVENT_LOG)
  finally:
    engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT, engine.ALTERNATE_STREAMS = None, 1, saved
  kinds = sorted((e['kind'], e.get('flags', 0) & 1) for e in log)
  # one pass over the members (per mask setting), nothing else
  assert kinds == [('ens_binned', 1 if with_mask else 0)], kinds
  if nan_mask:  # the per-point route: the mask is NOT on the W dims only
    from weatherbenchx_amd import _hip as _h
    assert not (log[0]['w_flags'] & _h.BINNED_MASK_ON_W), log
  points = nl * env.nlat * env.nlon
  k_ms = float(np.mean([e['ms'] for e in log]))
  name = f"ens_atoms_kernel<{m},true,{'NT' if env.layout == 'lon_fastest' else 'L2-shared lines'}> behind wbx_ens_binned"
  roof = kernel_roofline(name, k_ms, points * ((m + 1) * 4 + (1 if nan_mask else 0)),
                         pmc_traffic('ens_atoms_kernel', not args.small,
                                     f"public_chunk_ens{'_ifs' if ifs_layout else ''}{'_nanmask' if nan_mask else ''}@{env.layout}"))
  roof['kernel_ms_source'] = 'HIP events around 10 back-to-back repetitions of each launch of a chunk, mean per launch (one stream)'
  roof['launch_includes'] = ('aid_merge (mask folded into the atom ids, masked launch only; per point of the chunk for the NaN mask) + '
                             'ens_atoms_kernel with its in-kernel sums over patches: no second-stage or finish kernel')
  if nan_mask:
    roof['bytes_per_point'] = (m + 1) * 4 + 1
  # a sampled check against the oracle, outside the timed region: one lead time, every bin of CRPS
  from oracle import wbx_oracle as O
  lead = nl - 1
  sel_p = ens[0, :, lead] if ifs_layout else ens[0, lead]
  pv = sel_p.cpu().numpy()
  tvh = tv[0, lead].cpu().numpy()
  od_p, od_t = ('number',) + env.sp, env.sp
  skill, sd = O.crps_skill(pv, od_p, tvh, od_t, 'number')
  spread, _ = O.crps_spread(pv, od_p, 'number', fair=True, use_sort=True)
  names, masks = O.region_masks(env.lat, env.lon, REGIONS, land_sea_mask=land)
  bm = [('region', masks, ('region', 'latitude', 'longitude'))]
  w = (O.grid_area_weights(env.lat), ('latitude',))
  use_valid = (nan_valid[lead] if nan_mask else valid) if with_mask else None
  a = O.aggregate(skill, sd, ['latitude', 'longitude'], weights=[w], bin_masks=bm, mask=use_valid,
                  mask_dims=('latitude', 'longitude') if with_mask else None)
  b = O.aggregate(spread, sd, ['latitude', 'longitude'], weights=[w], bin_masks=bm)
  want = O.crps(a[0] / a[1], b[0] / b[1])
  got = np.asarray(out['crps.v'].transpose('lead_time', 'region').values)[lead]
  ok = np.isfinite(want)
  err = float(np.max(np.abs(got[ok] / want[ok] - 1.0)))
  assert list(out['crps.v']['region'].values) == names and err < 1e-6, err
  def make_chunk(i, ic):
    cs = dict(coords, init_time=ic)
    p = xr.DataArray(ens, dims=pdims, coords={k: v for k, v in cs.items() if k in pdims})
    t = xr.DataArray(tv, dims=tdims, coords={k: v for k, v in cs.items() if k in tdims})
    if nan_mask:
      t = t.assign_coords(mask=nan_mask_da)
    elif with_mask:
      t = t.assign_coords(mask=mask_da)
    return {'v': p}, {'v': t}
  loop_ms, loop_stats, _ = chunk_loop_ms(env, make_chunk, metrics, agg, coords['lead_time'], int(os.environ.get('WBX_BENCH_CHUNKS', 120 if not args.small else 12)))
  del ens, tv
  return {'workload': f"public benchmark chunk, probabilistic: f32[1 init,{nl} lead,{m} member,{env.nlat},{env.nlon}] "
                      f"({'init,number,lead' if ifs_layout else 'init,lead,number'} order) vs f32[1,{nl},{env.nlat},{env.nlon}] "
                      f"{('with NaN targets + add_nan_mask_to_data (per-point mask, another hole per lead)' if nan_mask else 'with a (latitude,longitude) mask coordinate') if with_mask else 'without a mask coordinate'}, CRPS(fair) + unbiased spread/skill + unbiased-mean RMSE + mean RMSE, "
                      f'GridAreaWeighting, {len(REGIONS)} regions x land/sea = {2 * len(REGIONS)} bins, masked=True, {env.layout}',
          'ms_per_chunk': loop_ms, 'value': points * len(metrics) / (loop_ms * 1e-3), 'unit': 'evals/s',
          'ms_per_chunk_is': 'pipeline.evaluate_chunks over 120 such chunks (sums in HBM, values at the end; chunk records on, '
                             'consecutive chunks on alternating launch streams)',
          'chunk_over_kernel': round(loop_ms / k_ms, 3), 'chunk_records': loop_stats, 'ms_per_chunk_values_every_chunk': ms_chunk,
          'launches_per_chunk': len(log), 'kernels': [e['kind'] for e in log],
          'ms_per_launch': [round(e['ms'], 4) for e in log], 'roofline': roof,
          'check': {'crps_global': float(got[0]), 'max_rel_err_vs_oracle_crps_all_bins_one_lead': err}}


# ---- configs[3] as BASELINE.json words it: spectra + the full deterministic suite in the same sweep -------------------------
def configs3_composite(env, nlead, nlev):
  """z f32[1 init, nlead, nlev, lat, lon] p, t + climatology -> RMSE / MSE / MAE / bias / ACC / activity per (lead, level) AND the
  zonal spectra of p and t per (lead, level), as two evaluations that share their loader through pipeline.evaluate_passes:
  on longitude-fastest 1440-point rows ONE kernel (wbx_det_spectrum) reads p, t, c once for both."""
  from weatherbenchx_amd import aggregation, engine, pipeline, spectra, time_chunks, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import deterministic
  args = env.args
  # (a job's set-up and read-back are paid once: 80 chunks per job by default -- configs[4] has 366 -- so that a chunk carries 1/80 of them)
  nchunk = max(8, args.steps * 8) if not args.small else 4
  lead_time = (np.arange(nlead) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(nchunk) * np.timedelta64(24, 'h')
  level = np.arange(nlev)
  zdims = ('init_time', 'lead_time', 'level') + env.sp
  cdims = ('dayofyear', 'hour', 'level') + env.sp
  sp_shape = env.sp_shape()
  pool = [(env.randn((1, nlead, nlev) + sp_shape, 280.0), env.randn((1, nlead, nlev) + sp_shape, 280.0)) for _ in range(2)]
  ndoy = 8
  clim = xr.Dataset({'z': xr.DataArray(env.randn((ndoy, 4, nlev) + sp_shape, 280.0, 10.0), dims=cdims, coords={
      'dayofyear': np.arange(1, ndoy + 1), 'hour': np.array([0, 6, 12, 18]), 'level': level, 'latitude': env.lat, 'longitude': env.lon})})
  env.torch.cuda.synchronize()
  index_of = {int(t.astype('int64')): i for i, t in enumerate(init_times)}
  ring = np.datetime64('2020-01-01T00', 'ns') + np.arange(4) * np.timedelta64(24, 'h')

  def load(inits, leads):
    i = index_of[int(inits[0].astype('int64'))]
    cs = {'init_time': inits, 'lead_time': lead_time, 'level': level, 'latitude': env.lat, 'longitude': env.lon,
          'valid_time': (('init_time', 'lead_time'), ring[i % 4] + lead_time[None, :])}
    return {'z': xr.DataArray(pool[i % 2][0], dims=zdims, coords=cs)}, {'z': xr.DataArray(pool[i % 2][1], dims=zdims, coords=cs)}
  det = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias(),
         'acc': deterministic.ACC(clim), 'prediction_activity': deterministic.PredictionActivity(clim)}
  spec = {'spectrum_p': spectra.ZonalPowerSpectrum('predictions'), 'spectrum_t': spectra.ZonalPowerSpectrum('targets')}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  passes = [('deterministic', load, det, area), ('spectra', load, spec, zonal)]

  def run(times):
    st = pipeline.evaluate_passes(times, passes)
    return st['deterministic'][None].metric_values(det), st['spectra'][None].metric_values(spec)
  run(time_chunks.TimeChunks(init_times[:3], lead_time, init_time_chunk_size=1))
  env.sync()
  # a job of this size is ~60 ms: nine of them.  The first still builds launch plans for time labels the warm-up did not see and
  # runs on ramping clocks (2.2-2.7 ms per chunk against 0.95-1.0 for the others), and inside the full bench line (not when
  # this leg runs alone) one or two further jobs take 1.5-2.1 ms: the median of the last eight is reported, every job is listed
  runs = []
  for _ in range(max(2, int(os.environ.get('WBX_BENCH_COMPOSITE_RUNS', '9')))):
    t0 = time.perf_counter()
    dvals, svals = run(time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1))
    env.sync()
    runs.append((time.perf_counter() - t0) / nchunk * 1e3)
  ms = float(np.median(runs[1:]))
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 5
  run(time_chunks.TimeChunks(init_times[:2], lead_time, init_time_chunk_size=1))
  log = list(engine.S1_EVENT_LOG)
  engine.S1_EVENT_LOG = None
  points = nlead * nlev * env.nlat * env.nlon
  fused = [e for e in log if e['kind'] == 'det_spectrum']
  out = {'workload': f'configs[3]: z f32[1,{nlead},{nlev},{env.nlat},{env.nlon}] p, t + climatology per chunk -> rmse/mse/mae/bias/acc/'
                     f'activity AND zonal power spectra of p and t per (lead, level), {env.layout}; two evaluations sharing a loader '
                     '(pipeline.evaluate_passes)',
         'chunks': nchunk, 'ms_per_chunk': ms, 'ms_per_chunk_runs': [round(r, 4) for r in runs],
         'value': points * (len(det) + len(spec)) / (ms * 1e-3), 'unit': 'evals/s',
         'algorithmic_bytes_per_point': 12, 'frac_of_hbm_peak': round(points * 12 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         'launches_per_chunk': {k: sum(1 for e in log if e['kind'] == k) // 2 for k in sorted({e['kind'] for e in log})},
         'check': {'rmse_mean': float(np.asarray(dvals['rmse.z'].values).mean()),
                   'sum_k_S_k': float(np.asarray(svals['spectrum_p.z'].values)[0, 0].sum()), 'expected': 280.0 ** 2 + 1.0}}
  if fused:
    kname = 'zspec1440_det_latfast_kernel' if fused[0].get('slab_rows') else 'zspec1440_det_kernel'
    out['roofline'] = kernel_roofline(kname + '<true> (spectra of p and t + DET6 lanes, one sweep; + 2 x 5 us memsets)',
                                      float(np.median([e['ms'] for e in fused])), points * 12,
                                      pmc_traffic(kname, not args.small, f'spectrum@{env.layout}'))
  del pool
  return out


# ---- zonal spectra ----------------------------------------------------------------------------------------------------
def spectrum_leg(env):
  from weatherbenchx_amd import aggregation, engine, spectra, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  args = env.args
  nt_s, nlev_s = (8, 37) if not args.small else (2, 3)
  sdims = ('lead_time', 'level') + env.sp
  scoords = {'latitude': env.lat, 'longitude': env.lon}
  shp = (nt_s, nlev_s) + env.sp_shape()
  sp_p = {'z': xr.DataArray(env.randn(shp, 280.0), dims=sdims, coords=scoords)}
  sp_t = {'z': xr.DataArray(env.randn(shp, 280.0), dims=sdims, coords=scoords)}
  env.torch.cuda.synchronize()
  smetrics = {'spectrum_p': spectra.ZonalPowerSpectrum('predictions'), 'spectrum_t': spectra.ZonalPowerSpectrum('targets')}
  sagg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])

  def launch():
    return sagg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(smetrics, fresh(sp_p), fresh(sp_t)))

  def srun(n):  # the spectrum read-back goes through the deferred-result path like the other legs
    with engine.deferred_results():
      return pipelined(launch, lambda s: s.metric_values(smetrics), n)
  srun(3)
  env.sync()
  nsteps = args.steps * 5  # (a step is ~0.6 ms: the pipeline's fill and drain would be a tenth of ten steps)
  t0 = time.perf_counter()
  sout = srun(nsteps)
  env.sync()
  s_ms = (time.perf_counter() - t0) / nsteps * 1e3
  spoints = nt_s * nlev_s * env.nlat * env.nlon
  parseval = float(np.asarray(sout['spectrum_p.z'].values)[0].sum())
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 10  # the kernel alone: HIP events, 10 launches per pair
  for _ in range(2):
    launch().metric_values(smetrics)
  slog = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'spectrum']
  engine.S1_EVENT_LOG = None
  sk_ms = float(np.mean([e['ms'] for e in slog]))
  if env.nlon == 1440:
    kname = ('zspec1440_kernel (one wave per row pair, 720 = 12 x 5 x 12)' if env.layout == 'lon_fastest' else
             'zspec1440_latfast_kernel (24 adjacent rows per block step, staged through the LDS)')
  else:
    kname = 'zspec_fused_kernel'
  both = configs3_composite(env, nt_s, nlev_s)
  return {'with_deterministic_suite': both,
          'workload': f'configs[3]: zonal power spectra of p and t, f32[{nt_s},{nlev_s},{env.nlat},{env.nlon}] each, area-weighted '
                      f'mean over (lead_time, latitude), {env.layout}; fused in-LDS FFT + fp64 |F|^2 reduction (one pass over '
                      'the field); parity unpinned (no reference implementation, SURVEY F3)',
          'value': spoints * 2 / (s_ms * 1e-3), 'unit': 'field-points/s', 'ms_per_step': s_ms, 'steps': nsteps,
          'algorithmic_GBps': round(spoints * 2 * 4 / (s_ms * 1e-3) / 1e9, 1),
          'frac_of_hbm_peak': round(spoints * 2 * 4 / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
          'roofline': kernel_roofline(kname + ' (+ 6 us memset of the output), one field', sk_ms, spoints * 4,
                                      pmc_traffic(kname, not args.small, f'spectrum@{env.layout}')),
          'check': {'sum_k_S_k': parseval, 'expected': 280.0 ** 2 + 1.0}}


def distributed_shard(times, env):
  from weatherbenchx_amd import distributed
  return distributed.shard_chunks(list(times.iter_with_chunk_offsets()), env.rank, env.world)


def _lat_archive(env, pool, nlead, nlev, lead_time):
  """Writes the z fields of `pool` as a latitude-fastest archive -- p [init, lead, level, longitude, latitude], t [time, level,
  longitude, latitude] .npy files in /dev/shm --, reads every chunk back through `loaders.*FromFiles(device_layout=
  'lon_fastest')` (the copy into page-locked memory is wbx_host_transpose) and puts what arrived into the pool.  -> what the
  host paid: transposition against the plain copy of the same bytes, and the check that what arrived is the archive transposed."""
  import shutil
  import tempfile
  from weatherbenchx_amd import loaders
  torch = env.torch
  npool = len(pool)
  root = '/dev/shm' if os.path.isdir('/dev/shm') else None
  d = tempfile.mkdtemp(prefix='wbx_lat_archive_', dir=root)
  threads = max(1, min(16, (os.cpu_count() or 2) // 2))
  try:
    inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(npool) * np.timedelta64(24, 'h')
    ntime = (npool - 1) * 4 + nlead  # valid times of daily inits with 6-hourly leads
    times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ntime) * np.timedelta64(6, 'h')
    pp, tp = os.path.join(d, 'p.npy'), os.path.join(d, 't.npy')
    pm = np.lib.format.open_memmap(pp, mode='w+', dtype=np.float32, shape=(npool, nlead, nlev, env.nlon, env.nlat))
    tm = np.lib.format.open_memmap(tp, mode='w+', dtype=np.float32, shape=(ntime, nlev, env.nlon, env.nlat))
    for k, buf in enumerate(pool):  # (archive order = the pool's fields transposed; t rows that two inits share: the later one's)
      pm[k] = buf['z_p'][0].transpose(-1, -2).contiguous().cpu().numpy()
      tm[4 * k:4 * k + nlead] = buf['z_t'][0].transpose(-1, -2).contiguous().cpu().numpy()
    pm.flush()
    tm.flush()
    del pm, tm
    dims = ('level', 'longitude', 'latitude')
    coords = {'level': np.arange(nlev), 'longitude': env.lon, 'latitude': env.lat}
    out = {'files': {'p': [npool, nlead, nlev, env.nlon, env.nlat], 't': [ntime, nlev, env.nlon, env.nlat], 'dir': root or 'tmp'},
           'loader_threads': threads}
    for name, layout in (('plain_copy', None), ('transposed', 'lon_fastest')):
      lp = loaders.PredictionsFromFiles({'z': pp}, inits, lead_time, dims, coords, device_layout=layout, threads=threads)
      lt = loaders.TargetsFromFiles({'z': tp}, times, dims, coords, device_layout=layout, threads=threads)
      for rep in range(2):  # (the second round reads from a warm page cache into warm page-locked blocks)
        lp.timings.update(bytes=0, seconds=0.0)
        lt.timings.update(bytes=0, seconds=0.0)
        got = [(lp.load_chunk(inits[k:k + 1], lead_time)['z'], lt.load_chunk(inits[k:k + 1], lead_time)['z']) for k in range(npool)]
      nbytes, secs = lp.timings['bytes'] + lt.timings['bytes'], lp.timings['seconds'] + lt.timings['seconds']
      out[name + '_GBps'] = round(nbytes / secs / 1e9, 2)
      out[name + '_ms_per_chunk'] = round(secs / npool * 1e3, 1)
      if layout is not None:
        ok = True
        for k, (pc, tc_) in enumerate(got):
          assert pc.dims[-2:] == ('latitude', 'longitude') and tc_.dims[-2:] == ('latitude', 'longitude')
          zp = torch.as_tensor(np.asarray(pc.data)).to(env.dev)
          zt = torch.as_tensor(np.asarray(tc_.data)).to(env.dev)
          ok = ok and bool(torch.equal(zp, pool[k]['z_p'])) and (k + 1 < npool or bool(torch.equal(zt, pool[k]['z_t'])))
          pool[k]['z_p'], pool[k]['z_t'] = zp, zt
        out['arrived_equals_archive_transposed'] = ok
      del got
    out['transposition_over_plain_copy'] = round(out['plain_copy_GBps'] / out['transposed_GBps'], 2)
    out['note'] = ('the loader threads copy every chunk from the page cache into page-locked memory either way; with device_layout= '
                   "'lon_fastest' that copy is the blocked transposition.  A file-backed job is bound by this copy and the H2D "
                   'behind it, not by its kernels: the chunk loop below is timed on what arrived, resident')
    return out
  finally:
    shutil.rmtree(d, ignore_errors=True)


# ---- configs[4]: the full suite streamed over (init x lead) chunks, sharded over the ranks ------------------------
def config5_leg(env, lat_archive=False):
  """`lat_archive`: z and its climatology come from a LATITUDE-FASTEST archive ([.., longitude, latitude] .npy files in
  /dev/shm, the layout of the public stores) through the transposing loader / slab pool (`device_layout='lon_fastest'`): the
  device sees longitude-fastest fields and runs the same kernels as the plain leg; timed with every slab resident."""
  from weatherbenchx_amd import aggregation, pipeline, spectra, time_chunks, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import deterministic
  args = env.args
  ninit = args.config5_inits if not args.small else 6
  nlead, nlev, m = (20, 37, 51) if not args.small else (3, 2, 5)
  npool = 2
  lead_time = (np.arange(nlead) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ninit) * np.timedelta64(24, 'h')
  level = np.arange(nlev)
  zdims = ('init_time', 'lead_time', 'level') + env.sp
  edims = ('init_time', 'lead_time', 'number') + env.sp
  tdims = ('init_time', 'lead_time') + env.sp
  cdims = ('dayofyear', 'hour', 'level') + env.sp
  sp_shape = env.sp_shape()
  # resident pool of FIELDS (their H2D is excluded and stated): chunk i reads buffer i mod npool (its own generator, the same on
  # every rank: chunk i holds the same numbers whichever rank runs it, so the result does not depend on N).
  # The CLIMATOLOGY is a whole calendar -- [366 dayofyear, 4 hours, level, lat, lon], 225 GB at 37 levels -- in host memory behind
  # a slab pool (climatology_cache.py): a chunk's valid times name 20 of its 1464 (dayofyear, hour) slabs, daily inits share 16
  # of them with their predecessor, the 4 new ones are uploaded one chunk ahead on a copy stream.  (Host memory: the calendar is
  # a view whose dayofyear stride is 0 over ONE page-locked day -- generating 225 GB would take longer than the whole bench; the
  # pool sees 1464 distinct slabs all the same: a slab is keyed by its position and uploaded when it is missed.)
  from weatherbenchx_amd import climatology_cache
  gen5 = env.torch.Generator(device=env.dev)
  gen5.manual_seed(4242)
  pool = []
  for _ in range(npool):
    ens = env.randn((1, nlead, m) + sp_shape, gen=gen5)
    t2 = env.randn((1, nlead) + sp_shape, 280.0, gen=gen5)
    ens += t2[:, :, None]
    t2 += env.randn((1, nlead) + sp_shape, gen=gen5)
    pool.append({'z_p': env.randn((1, nlead, nlev) + sp_shape, 280.0, gen=gen5),
                 'z_t': env.randn((1, nlead, nlev) + sp_shape, 280.0, gen=gen5), 't2m_p': ens, 't2m_t': t2})
  ndoy = 366
  archive = None
  c_shape, c_dims = (4, nlev) + sp_shape, cdims
  if lat_archive:
    archive = _lat_archive(env, pool, nlead, nlev, lead_time)  # z_p / z_t of the pool now are what the loaders delivered
    c_shape, c_dims = (4, nlev, env.nlon, env.nlat), ('dayofyear', 'hour', 'level', 'longitude', 'latitude')
  day = env.ctx.pinned_empty(c_shape, np.float32)
  day[...] = env.randn(c_shape, 280.0, 10.0, gen=gen5).cpu().numpy()
  clim_slots = 48  # two chunks' worth of slabs and some: 7.4 GB at 37 levels
  clim = climatology_cache.cached(xr.Dataset({'z': xr.DataArray(np.broadcast_to(day[None], (ndoy,) + day.shape), dims=c_dims, coords={
      'dayofyear': np.arange(1, ndoy + 1), 'hour': np.array([0, 6, 12, 18]), 'level': level,
      'latitude': env.lat, 'longitude': env.lon})}), slots=clim_slots, device_layout='lon_fastest' if lat_archive else None,
      threads=16 if lat_archive else 4)
  slab_bytes = int(np.prod((nlev,) + sp_shape)) * 4
  env.torch.cuda.synchronize()
  index_of = {int(t.astype('int64')): i for i, t in enumerate(init_times)}

  def coords_for(inits):
    # (real labels: the climatology is addressed at dayofyear / hour of init_time + lead_time)
    i = index_of[int(inits[0].astype('int64'))]
    return i, {'init_time': inits, 'lead_time': lead_time, 'latitude': env.lat, 'longitude': env.lon}

  def load_det(inits, leads):
    i, cs = coords_for(inits)
    buf = pool[i % npool]
    cz = dict(cs, level=level)
    return ({'z': xr.DataArray(buf['z_p'], dims=zdims, coords=cz)}, {'z': xr.DataArray(buf['z_t'], dims=zdims, coords=cz)})

  def load_ens(inits, leads):
    i, cs = coords_for(inits)
    buf = pool[i % npool]
    return ({'t2m': xr.DataArray(buf['t2m_p'], dims=edims, coords=cs)}, {'t2m': xr.DataArray(buf['t2m_t'], dims=tdims, coords=cs)})

  det = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias(),
         'acc': deterministic.ACC(clim), 'prediction_activity': deterministic.PredictionActivity(clim)}
  spec = {'spectrum_p': spectra.ZonalPowerSpectrum('predictions'), 'spectrum_t': spectra.ZonalPowerSpectrum('targets')}
  ens = ensemble_suite()
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  passes = [('deterministic', load_det, det, area), ('spectra', load_det, spec, zonal), ('ensemble', load_ens, ens, area)]
  metrics_of = {name: m for name, _, m, _ in passes}
  # ONE job: the chunks of the three evaluations run interleaved, all their accumulators live in one engine.Accumulation and
  # cross the ranks in ONE all-reduce at the end (pipeline.evaluate_passes; beam_pipeline.py:509-510 combines every key of
  # the job in one CombinePerKey) -- through the library's own RCCL entry point (wbx_acc_allreduce) unless WBX_COLLECTIVE=torch.
  comm = env.comm
  stats = {}

  def run(times):
    states = pipeline.evaluate_passes(times, passes, rank=env.rank, world_size=env.world, comm=comm, stats=stats)
    return {name: states[name][None].metric_values(metrics_of[name]) for name in metrics_of}  # (the sums are on the host here)

  def run_some(times, names):  # a subset of the evaluations as their own job, for the per-evaluation timings below
    some = [p for p in passes if p[0] in names]
    tp = time.perf_counter()
    st = pipeline.evaluate_passes(times, some, rank=env.rank, world_size=env.world, all_reduce=False)
    for name, _, metrics, _ in some:
      st[name][None].metric_values(metrics)
    return time.perf_counter() - tp
  cache = climatology_cache.cache_for(clim['z'])
  warm = time_chunks.TimeChunks(init_times[:2 * env.world], lead_time, init_time_chunk_size=1)
  run(warm)
  env.sync()
  times = time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1)
  from weatherbenchx_amd import replay as _replay
  _replay.reset_stats()
  ring_inits = np.tile(init_times[:6], ninit // 6 + 1)[:ninit]
  if lat_archive:
    # (the year-long stream is the plain leg's; here: inits cycling through six days, every slab in the pool after one round)
    times = time_chunks.TimeChunks(ring_inits, lead_time, init_time_chunk_size=1)
    run(time_chunks.TimeChunks(ring_inits[:12 * env.world], lead_time, init_time_chunk_size=1))
    env.sync()
    _replay.reset_stats()
  c0 = dict(cache.stats)
  t0 = time.perf_counter()
  out = run(times)
  env.sync()
  rank_s = time.perf_counter() - t0
  dt = env.max_over_ranks(rank_s)
  c1 = dict(cache.stats)
  record_stats = {k: v for k, v in _replay.STATS.items() if k != 'refusals'}
  record_stats['refusals'] = [str(x)[:160] for x in _replay.STATS['refusals'][:2]]
  my_chunks = max(len(distributed_shard(times, env)), 1)
  up_bytes = c1['upload_bytes'] - c0['upload_bytes']
  clim_stats = {'host_shape': [ndoy, 4, nlev] + list(sp_shape), 'host_GB': round(ndoy * 4 * slab_bytes / 1e9, 1),
                'slab_MB': round(slab_bytes / 1e6, 1), 'pool_slots': cache.nslots, 'pool_GB': round(cache.nslots * slab_bytes / 1e9, 2),
                'uploads': c1['uploads'] - c0['uploads'], 'evictions': c1['evictions'] - c0['evictions'],
                'asked_one_chunk_ahead': c1['prefetched'] - c0['prefetched'],
                'h2d_MB_per_chunk_rank0': round(up_bytes / my_chunks / 1e6, 1),
                'h2d_GBps_rank0': round(up_bytes / rank_s / 1e9, 1),
                'note': ('every slab of the timed loop is in the pool (inits cycling through six days)' if lat_archive else
                         'the timed job is bound by this stream (PCIe), not by its kernels: see all_slabs_resident')}
  # the same loop with every slab it names already in the pool (inits cycling through six days: 40 slabs, 48 slots): what the
  # kernels and the host cost per chunk when the copy stream is idle -- the gather tables are pool slots either way
  hits = None
  if not lat_archive:
    ring = time_chunks.TimeChunks(ring_inits, lead_time, init_time_chunk_size=1)
    run(time_chunks.TimeChunks(ring_inits[:12 * env.world], lead_time, init_time_chunk_size=1))
    env.sync()
    h0 = dict(cache.stats)
    t0 = time.perf_counter()
    run(ring)
    env.sync()
    hits_dt = env.max_over_ranks(time.perf_counter() - t0)
    hits = (hits_dt, cache.stats['uploads'] - h0['uploads'])
  # per-pass pace of this rank (outside the timed region, a sixth of the chunks, no collective)
  sub = time_chunks.TimeChunks(ring_inits[:max(2 * env.world, ninit // 6)], lead_time, init_time_chunk_size=1)
  nsub = len(distributed_shard(sub, env))
  from weatherbenchx_amd import engine as _engine
  fused = _engine.FUSE_DET_SPECTRA and env.nlon == 1440 and (env.layout == 'lon_fastest' or _engine.FUSE_DET_SPECTRA_LATFAST)
  pass_s = {'z: deterministic + spectra of p and t' + (' (ONE sweep: wbx_det_spectrum)' if fused else ' (separate launches)'):
                run_some(sub, ('deterministic', 'spectra')) / max(nsub, 1) * 1e3,
            'deterministic alone': run_some(sub, ('deterministic',)) / max(nsub, 1) * 1e3,
            'spectra alone': run_some(sub, ('spectra',)) / max(nsub, 1) * 1e3,
            'ensemble': run_some(sub, ('ensemble',)) / max(nsub, 1) * 1e3}
  env.sync()
  grid = env.nlat * env.nlon
  pz, pt = nlead * nlev * grid, nlead * grid
  evals_per_chunk = pz * len(det) + pz * len(spec) + pt * len(ens)
  # one sweep over p, t, c serves the deterministic lanes AND both spectra (12 B/point); as separate launches p and t are read
  # again by the spectra (20 B/point of traffic for the same 12 B/point of algorithmic input -- the fraction below counts 12)
  bytes_per_chunk = pz * 12 + pt * (m + 1) * 4
  rm = float(np.asarray(out['deterministic']['rmse.z'].values).mean())
  return {'workload': ('z and its climatology from a latitude-fastest archive through the transposing loader / slab pool; ' if lat_archive else '') +
                      f'configs[4]: full suite on {ninit} inits x {nlead} leads, streamed as [1 init x {nlead} lead] chunks from a '
                      f'resident pool of {npool} (H2D excluded): z f32[{nlead},{nlev},{env.nlat},{env.nlon}] p,t + a [366,4,{nlev},..] host '
                      f'climatology behind a {clim_slots}-slab device pool (its H2D INCLUDED) -> '
                      f'rmse/mse/mae/bias/acc/activity + zonal spectra of p and t; t2m f32[{nlead},{m},{env.nlat},{env.nlon}] -> '
                      'crps/spread-skill/unbiased-mean rmse/mean rmse; reduce (init_time, latitude, longitude) -> per (lead, level)',
          'sharding': f'contiguous runs of chunks per rank ({env.world}: consecutive inits share climatology slabs); the three evaluations interleaved chunk by chunk (pipeline.evaluate_passes), '
                      'every accumulator in HBM, ONE all-reduce for the whole job at the end',
          'collectives': stats.get('collectives'), 'accumulator_values': stats.get('accumulator_values'),
          'collective_backend': env.collective, 'collective': _collective_record(env),
          'rccl_ranks': env.world if (env.world > 1 and args.backend == 'nccl') else 0,
          'scaling': 'strong', 'n_gpus': env.world, 'chunks': ninit, 'time_slices': ninit * nlead, 'seconds': dt,
          'seconds_rank0': rank_s, 'ms_per_chunk': dt / ninit * 1e3, 'ms_per_chunk_rank0': rank_s / max(len(distributed_shard(times, env)), 1) * 1e3,
          'ms_per_chunk_by_pass_rank0': {k: round(v, 3) for k, v in pass_s.items()},
          'ms_per_chunk_by_pass_note': 'subsets of the evaluations as their own jobs on a sixth of the chunks, outside the timed region',
          'z_bytes_per_point': 12, 'z_traffic_bytes_per_point': 12 if fused else 20, 'fused_det_spectra': bool(fused),
          'chunk_records': record_stats,
          'climatology': clim_stats,
          'all_slabs_resident': None if hits is None else {
              'ms_per_chunk': hits[0] / ninit * 1e3, 'uploads_in_the_timed_loop': hits[1],
              'value': evals_per_chunk * ninit / hits[0], 'unit': 'evals/s',
              'frac_of_hbm_peak_per_gpu': round(bytes_per_chunk * ninit / hits[0] / 1e9 / HBM_PEAK_GBS / env.world, 4),
              'note': 'inits cycling through six days: every slab of the loop is in the pool; kernels + host per chunk'},
          'archive': archive,
          'value': evals_per_chunk * ninit / dt, 'unit': 'evals/s',
          'algorithmic_GBps': round(bytes_per_chunk * ninit / dt / 1e9, 1),
          'frac_of_hbm_peak_per_gpu': round(bytes_per_chunk * ninit / dt / 1e9 / HBM_PEAK_GBS / env.world, 4),
          'check': {'rmse_z_mean': rm, 'crps_t2m_mean': float(np.asarray(out['ensemble']['crps.t2m'].values).mean()),
                    'spectrum_p_z_mean': float(np.asarray(out['spectra']['spectrum_p.z'].values).mean()),
                    'shape_rmse_z': list(out['deterministic']['rmse.z'].shape)}}


# ---- CPU baseline ---------------------------------------------------------------------------------------------------
def cpu_leg(env, keep):
  """The oracle's reference-structure NumPy path on bounded samples, on this box's host cores: the ensemble suite of the main
  line first (top-level `value`), the deterministic suite of configs1 beside it; each on one process and on all cores."""
  from oracle import cpu_workers
  from oracle import wbx_oracle as O
  args = env.args
  nworkers = args.cpu_workers or (os.cpu_count() or 1)
  w = O.grid_area_weights(env.lat)
  out = {}

  # -- ensemble suite: CRPS (rank form, fair) + variance + unbiased MSE + ensemble-mean SE on f32[51, lat, lon] fields
  m, nfield = 51, (8 if not args.small else 1)
  rng = np.random.default_rng(7)
  tv = rng.standard_normal((nfield, env.nlat, env.nlon), dtype=np.float32) + 280
  pv = tv[:, None] + rng.standard_normal((nfield, m, env.nlat, env.nlon), dtype=np.float32)
  tv = tv + rng.standard_normal((nfield, env.nlat, env.nlon), dtype=np.float32)
  t0 = time.perf_counter()
  crps = [float(r['CRPSSkill'] - 0.5 * r['CRPSSpread']) for r in (O.reference_structure_ensemble(pv[k], tv[k], w) for k in range(nfield))]
  cdt = time.perf_counter() - t0
  epoints = nfield * env.nlat * env.nlon
  n_emetrics = 4
  one = epoints * n_emetrics / cdt
  ens = {'value': one, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
         'sample': f'{nfield} of the 37 levels: f32[{m},{env.nlat},{env.nlon}] fields of the main line\'s workload ({epoints} points, '
                   f'{cdt:.1f} s; NumPy argsort ranks + elementwise + einsum, single thread; 4 metrics)',
         'check_crps': float(np.mean(crps))}
  del pv, tv
  try:
    band, per = (90, 2) if not args.small else (env.nlat, 1)  # a worker's field is a 90-latitude band: ~0.3 GB per process
    # the NumPy path is bound by memory bandwidth long before it runs out of cores (256 workers: 3.9 x one core in round 4): a
    # quarter of the logical CPUs is tried beside all of them, and the BETTER of the two is the multi-core baseline
    tried = []
    for nw in ([nworkers] if (args.small or args.cpu_workers) else sorted({nworkers, max(1, nworkers // 4)}, reverse=True)):
      t0 = time.perf_counter()
      res = cpu_workers.run(nw, per, m, band, env.nlon, kind='ensemble')
      tried.append({'value': res['points'] * n_emetrics / res['seconds'], 'unit': 'evals/s', 'cores': nw, 'kind': 'port',
                    'sample': f"{nw} worker processes x {per} bands of f32[{m},{band},{env.nlon}] ({res['points']} points, "
                              f"{res['seconds']:.2f} s between the start barrier and the last finish; "
                              f'{time.perf_counter() - t0:.1f} s with process start-up)',
                    'speedup_over_one_core': res['points'] * n_emetrics / res['seconds'] / one})
    ens['all_cores'] = dict(max(tried, key=lambda r: r['value']), tried={r['cores']: round(r['value'], 1) for r in tried})
  except Exception as e:  # pylint: disable=broad-except
    ens['all_cores'] = {'value': None, 'error': f'{type(e).__name__}: {e}'}
  out.update({k: v for k, v in ens.items() if k != 'all_cores'})
  out['workload'] = 'ensemble suite of the main line (CRPS + unbiased spread/skill + unbiased-mean RMSE + ensemble-mean RMSE)'
  out['host_cpus'] = os.cpu_count()
  out['ensemble'] = ens

  # -- deterministic suite (configs1)
  if keep is not None:
    p_t, t_t, clim_t, _, nlev, ni, nl, n_metrics = keep
    si, sl = min(10, ni), min(5, nl)  # ~50 of the 400 (init, lead) slices: a few seconds of single-thread NumPy
    idx = (slice(0, si), slice(0, sl))
    ph, th = p_t[idx].cpu().numpy(), t_t[idx].cpu().numpy()
    # an already-aligned climatology of the same shape (the reference's .sel gather is NOT charged to the CPU side)
    ch = clim_t[:si, :1].expand(si, sl, *clim_t.shape[2:]).contiguous().cpu().numpy()
    if env.layout == 'lat_fastest':
      ph, th, ch = (np.ascontiguousarray(np.swapaxes(a, -1, -2)) for a in (ph, th, ch))
    t0 = time.perf_counter()
    O.reference_structure_deterministic(ph, th, ch, w)
    cdt = time.perf_counter() - t0
    spoints = int(np.prod(ph.shape))
    done = spoints * n_metrics / cdt
    det = {'value': done, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
           'sample': f'{si} init x {sl} lead x {nlev} level x {env.nlat} x {env.nlon} of configs1 '
                     f'({spoints} points, {cdt:.1f} s; NumPy elementwise + einsum, single thread; {n_metrics} metrics)'}
    try:
      per = 2 if not args.small else 1  # (init, lead) slices per worker
      t0 = time.perf_counter()
      res = cpu_workers.run(nworkers, per, nlev, env.nlat, env.nlon)
      det['all_cores'] = {'value': res['points'] * n_metrics / res['seconds'], 'unit': 'evals/s', 'cores': nworkers,
                          'kind': 'port', 'sample': f"{nworkers} worker processes x {per} (init, lead) slices x {nlev} level x "
                                                    f"{env.nlat} x {env.nlon} ({res['points']} points, {res['seconds']:.2f} s between the "
                                                    f'start barrier and the last finish; {time.perf_counter() - t0:.1f} s with process start-up)',
                          'speedup_over_one_core': res['points'] * n_metrics / res['seconds'] / done}
    except Exception as e:  # pylint: disable=broad-except
      det['all_cores'] = {'value': None, 'error': f'{type(e).__name__}: {e}'}
    out['deterministic'] = det
  return out


_JSON_FD = None


def _claim_stdout():
  """stdout carries the ONE JSON line and nothing else: file descriptor 1 is pointed at stderr for everything that prints on
  its own (RCCL's version banner, gloo's rank messages, torch warnings written from C++), and the line goes out through a
  private duplicate of the original descriptor."""
  global _JSON_FD
  if _JSON_FD is None:
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)


LINE_LIMIT = 6000  # bytes; the driver's parser lost the 20.9 KB line of round 4 (BENCH_r04.json: parsed null), 8 KB is its tail


def _short(s, n):
  s = str(s)
  return s if len(s) <= n else s[:n - 1] + '…'


def _sig(v, digits=5):
  return float(f'{float(v):.{digits}g}') if isinstance(v, (int, float)) and not isinstance(v, bool) else v


def _leg_record(leg):
  """One short record of a side leg: wall time per step / chunk, the kernel and its roofline fraction, counter traffic over
  algorithmic bytes, and the largest error against the oracle the leg measured (outside its timed region)."""
  roof = leg.get('roofline') if isinstance(leg.get('roofline'), dict) else (leg if 'frac' in leg and 'kernel_ms' in leg else {})
  rec = {}
  for k in ('ms_per_step', 'ms_per_chunk'):
    if isinstance(leg.get(k), (int, float)):
      rec['ms'] = _sig(leg[k], 4)
  if roof:
    rec['kernel'] = _short(str(roof.get('kernel', '')).split(' (')[0], 44)
    rec['kernel_ms'], rec['frac'] = _sig(roof.get('kernel_ms'), 4), _sig(roof.get('frac'), 4)
    if roof.get('traffic') and roof.get('algorithmic_bytes_per_launch'):
      rec['traffic_ratio'] = round(roof['traffic'] / roof['algorithmic_bytes_per_launch'], 3)
  elif isinstance(leg.get('frac_of_hbm_peak', leg.get('frac_of_hbm_peak_per_gpu')), (int, float)):
    rec['frac'] = _sig(leg.get('frac_of_hbm_peak', leg.get('frac_of_hbm_peak_per_gpu')), 4)
  if isinstance(leg.get('chunk_over_kernel'), (int, float)):
    rec['x_kernel'] = leg['chunk_over_kernel']  # ms per chunk of the chunk loop over the launch's kernel time
  if isinstance(leg.get('value'), (int, float)):
    rec['value'] = _sig(leg['value'], 4)
  errs = [v for k, v in (leg.get('check') or {}).items() if 'err' in k and isinstance(v, (int, float))]
  if errs:
    rec['oracle_err'] = _sig(max(errs), 2)
  return rec


_NOT_LEGS = ('roofline', 'check', 'config', 'cpu_baseline', 'legs', 'box', 'climatology', 'archive', 'chunk_records', 'collective')
_LEG_ALIASES = {'with_mask_coordinate': 'mask', 'with_nan_mask': 'nanmask', 'with_deterministic_suite': 'det', 'public_chunk_ens': 'pce',
                'public_chunk_ens_ifs_layout': 'pce_ifs', 'public_chunk': 'pc', 'lat_fastest': 'lat', 'default_crps_ensemble': 'default',
                'pairwise_form': 'pair', 'skipna_ensemble': 'skipna', 'all_slabs_resident': 'hits'}


def _collect_legs(node, prefix, out):
  for k, v in node.items():
    if not isinstance(v, dict) or k in _NOT_LEGS:
      continue
    name = (prefix + '.' if prefix else '') + _LEG_ALIASES.get(k, k)
    rec = _leg_record(v)
    if rec:
      out[name] = rec
    _collect_legs(v, name, out)


def compact_line(result, full_path=None):
  """The ONE line the driver parses: headline + roofline + cpu_baseline + a table of one short record per side leg.  Everything
  else (workload prose, per-leg roofline objects, checks) is in `bench_full.json` beside this script (path in the line)."""
  keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
          'dtype', 'data', 'skipped', 'reason', 'note')
  line = {k: result[k] for k in keep if k in result}
  cfg = result.get('config') or {}
  if cfg:
    line['config'] = {'workload': _short(cfg.get('workload', ''), 330)}
    line['config'].update({k: cfg[k] for k in ('points_per_step_per_gpu', 'members', 'metrics', 'layout', 'accumulators', 'sharding',
                                               'collectives_per_step', 'rccl_ranks', 'collective_backend', 'collective', 'prewarm') if k in cfg})
  roof = result.get('roofline') or {}
  if roof:
    line['roofline'] = {k: roof[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel_ms', 'kernel_ms_median',
                                             'kernel_ms_min_max', 'algorithmic_bytes_per_launch', 'bytes_per_point',
                                             'launches_per_step') if k in roof}
    line['roofline']['kernel'] = _short(roof.get('kernel', ''), 90)
    line['roofline']['kernel_ms_source'] = 'HIP events around every timed launch on its launch stream, mean'
    line['roofline']['traffic_source'] = 'rocprofv3 --pmc FETCH_SIZE(x2 gfx950)+WRITE_SIZE, separate passes (profiles/), same library md5'
  cpu = result.get('cpu_baseline') or {}
  if cpu:
    line['cpu_baseline'] = {k: (_short(cpu[k], 200) if k == 'sample' else cpu[k]) for k in ('value', 'unit', 'cores', 'kind', 'sample') if k in cpu}
    for fam in ('ensemble', 'deterministic'):
      ac = (cpu.get(fam) or {}).get('all_cores') or {}
      one = (cpu.get(fam) or {}).get('value')
      if fam == 'ensemble' and ac:
        line['cpu_baseline']['all_cores'] = {'value': _sig(ac.get('value')), 'cores': ac.get('cores')}
      elif one is not None:
        line['cpu_baseline'][fam] = {'value': _sig(one), 'cores': 1, 'all_cores_value': _sig(ac.get('value')), 'all_cores': ac.get('cores')}
  chk = result.get('check') or {}
  if chk:
    line['check'] = {k: _sig(chk[k], 6) for k in ('crps_mean', 'unbiased_spread_skill_mean', 'oracle_max_rel_err') if k in chk}
  if result.get('box'):
    line['box'] = result['box']
  legs = {}
  _collect_legs(result, '', legs)
  if legs:
    line['legs'] = legs
  if full_path:
    line['full'] = full_path
  text = json.dumps(line, separators=(',', ':'))
  # a line that outgrows the limit sheds detail in a fixed order instead of failing the parse: leg kernel names, then leg values
  for drop in ('kernel', 'value', 'kernel_ms'):
    if len(text.encode()) <= LINE_LIMIT:
      break
    for rec in legs.values():
      rec.pop(drop, None)
    text = json.dumps(line, separators=(',', ':'))
  assert len(text.encode()) < 8192, len(text.encode())
  return text


def _emit(result):
  """Full result -> bench_full.json (+ gpurun_out/ when present, + stderr); the compact line -> stdout."""
  full = json.dumps(result)
  name = 'bench_full.json' if result.get('n_gpus', 1) == 1 else f"bench_full_n{result.get('n_gpus')}.json"
  written = None
  for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
    try:
      if os.path.isdir(d):
        with open(os.path.join(d, name), 'w') as f:
          f.write(full + '\n')
        written = written or name
    except OSError:
      pass
  sys.stderr.write(full + '\n')
  sys.stderr.flush()
  sys.stdout.flush()
  os.write(_JSON_FD if _JSON_FD is not None else 1, (compact_line(result, written) + '\n').encode())


def main():
  args = parse()
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(self_launch(args))  # (the ranks inherit this process's stdout: rank 0 writes the line)
  _claim_stdout()
  if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")
  env = Env(args)
  legs = None if args.legs == 'all' else set(args.legs.split(','))
  want = lambda name: legs is None or name in legs
  result, keep = {'metric': METRIC, 'value': None, 'note': 'main leg not selected (--legs)'}, None
  if want('main'):
    result = main_leg(env)
  if not args.no_ens and env.world == 1:
    if want('configs1'):
      result['configs1'], keep = configs1_leg(env)
    if want('ensemble'):
      ns = args.ens_slices if not args.small else 2
      nvar = 6 if not args.small else 2
      name, leg = ens_leg(env, 'lead_time', ns, nvar, 'ensemble',
                          f'configs[2]: {nvar} vars x f32[{ns} slices,51 members,{env.nlat},{env.nlon}], CRPS(rank form, fair) + '
                          f'unbiased spread/skill + unbiased-mean RMSE + mean RMSE, {env.layout}')
      result[name] = leg
    if want('public_chunk'):
      result['public_chunk'] = public_chunk_leg(env)
    if want('public_chunk_ens'):
      result['public_chunk_ens'] = public_chunk_ens_leg(env, with_mask=False) if args.pce_mask in ('both', '0') else {}
      if args.pce_mask in ('both', '1'):
        result['public_chunk_ens']['with_mask_coordinate'] = public_chunk_ens_leg(env, with_mask=True)
      if args.pce_mask in ('both', 'nan'):
        result['public_chunk_ens']['with_nan_mask'] = public_chunk_ens_leg(env, with_mask='nan')
    if want('spectrum'):
      result['spectrum'] = spectrum_leg(env)
    if want('lat_fastest') and args.layout == 'lon_fastest':
      # the layout of the public archives (data_loaders/xarray_loaders.py:185-188): the same legs on [.., longitude, latitude]
      env.set_layout('lat_fastest')
      lf = {'note': 'the main line, configs1 and the spectrum leg again on latitude-fastest arrays (N = 1)'}
      m = main_leg(env)
      lf['main'] = {k: m[k] for k in ('value', 'unit', 'ms_per_step', 'roofline', 'check')}
      lf['main']['workload'] = m['config']['workload']
      c1, _ = configs1_leg(env)
      lf['configs1'] = c1
      lf['spectrum'] = spectrum_leg(env)
      if want('public_chunk_ens'):
        lf['public_chunk_ens'] = public_chunk_ens_leg(env, with_mask=False)
        lf['public_chunk_ens']['with_mask_coordinate'] = public_chunk_ens_leg(env, with_mask=True)
        lf['public_chunk_ens']['with_nan_mask'] = public_chunk_ens_leg(env, with_mask='nan')
        lf['public_chunk_ens_ifs_layout'] = public_chunk_ens_leg(env, ifs_layout=True, with_mask=False)
      env.set_layout('lon_fastest')
      if not args.no_config5 and want('config5'):
        lf['config5'] = config5_leg(env, lat_archive=True)
      result['lat_fastest'] = lf
  if not args.no_config5 and want('config5'):
    result['config5'] = config5_leg(env)
  if not args.no_cpu and env.world == 1 and env.rank == 0 and want('cpu'):
    result['cpu_baseline'] = cpu_leg(env, keep)
  if env.rank == 0:
    try:  # the clock this box sustains under load, beside the numbers it shaped (boxes of the pool differ by 4-8 %)
      result['box'] = {'shader_MHz_all_CUs_busy': round(env.ctx.clock_probe(2048), 1), 'shader_MHz_one_block': round(env.ctx.clock_probe(1), 1),
                       'host_cpus': os.cpu_count()}
    except Exception as e:  # pylint: disable=broad-except
      result['box'] = {'error': f'{type(e).__name__}: {e}'}
    _emit(result)
  if env.world > 1:
    env.dist.destroy_process_group()


if __name__ == '__main__':
  main()
