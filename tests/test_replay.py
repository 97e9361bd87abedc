"""Chunk records (weatherbenchx_amd/replay.py + wbx_chunk_replay): the steady state of the chunk loop as ONE library call per
chunk.  CPU part: the C entry point's argument checks (no device needed: every refused record fails before anything is
enqueued) and the refusal logic of the recorder.  GPU part (`-m gpu`): chunk loops with replay on and off give bit-identical
accumulators -- deterministic suite with a climatology (the gather table follows the chunk) under region bins and a mask, the
ensemble suite under bins with and without a mask coordinate, spectra fused into the deterministic sweep -- and a loop whose
inputs change shape falls back to the ordinary path."""
import ctypes as C

import numpy as np
import pytest

from weatherbenchx_amd import _hip
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import engine
from weatherbenchx_amd import pipeline
from weatherbenchx_amd import replay
from weatherbenchx_amd import spectra
from weatherbenchx_amd import time_chunks
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import deterministic, probabilistic, wrappers

REGIONS = {'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360)), 'nh': ((20, 90), (0, 360)),
           'sh': ((-90, -20), (0, 360)), 'europe': ((35, 75), (-12.5, 42.5)), 'namerica': ((25, 60), (240, 285))}


def _calls(entries):
  arr = (_hip.CallStruct * len(entries))()
  for i, (fn, args) in enumerate(entries):
    arr[i].fn, arr[i].nargs = fn, len(args)
    for j, v in enumerate(args):
      arr[i].args[j] = v
  return arr


def _replay(calls, relocs=(), slots=()):
  lib = _hip.load_library()
  rl = (_hip.RelocStruct * max(len(relocs), 1))()
  for i, (c, a, s, off) in enumerate(relocs):
    rl[i].call, rl[i].arg, rl[i].slot, rl[i].offset = c, a, s, off
  sl = (C.c_uint64 * max(len(slots), 1))(*slots)
  rc = lib.wbx_chunk_replay(C.addressof(calls), len(calls), C.addressof(rl), len(relocs), C.addressof(sl), len(slots))
  return rc, lib.wbx_last_error().decode()


def test_replay_entry_point_refuses_what_it_cannot_run():
  """No device needed: the record is checked call by call before anything is enqueued."""
  assert _replay(_calls([]))[0] == 0  # an empty record is a no-op
  rc, msg = _replay(_calls([(99, [0])]))
  assert rc == -1 and 'cannot be part of a record' in msg
  rc, msg = _replay(_calls([(_hip.FN_IDS['wbx_acc_add'], [0, 0, 0])]))  # wbx_acc_add takes five arguments
  assert rc == -1 and 'takes 5 arguments' in msg
  rc, msg = _replay(_calls([(_hip.FN_IDS['wbx_acc_add'], [0, 0, 0, 4, 0])]))  # dispatched: the callee refuses a NULL context
  assert rc == -1 and 'NULL' in msg
  rc, msg = _replay(_calls([(_hip.FN_IDS['wbx_acc_add'], [0, 0, 0, 4, 0])]), relocs=[(0, 7, 0, 0)], slots=[1])
  assert rc == -1 and 'out of range' in msg
  rc, msg = _replay(_calls([(_hip.FN_IDS['wbx_acc_add'], [0, 0, 0, 4, 0])]), relocs=[(0, 1, 3, 0)], slots=[1])
  assert rc == -1 and 'out of range' in msg
  # a relocation is applied before the call: acc = slots[0] + 16 reaches the callee (which still refuses the NULL context)
  calls = _calls([(_hip.FN_IDS['wbx_acc_add'], [0, 0, 0, 4, 0])])
  _replay(calls, relocs=[(0, 1, 0, 16)], slots=[4096])
  assert calls[0].args[1] == 4096 + 16


def test_fn_ids_match_the_header():
  import os
  import re
  _hip.load_library()
  header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'wbx.h')).read()
  enum = dict(re.findall(r'WBX_FN_([A-Z0-9_]+) = (\d+)', header))
  assert {f'wbx_{k.lower()}': int(v) for k, v in enum.items()} == _hip.FN_IDS
  assert int(re.search(r'#define WBX_CALL_MAX_ARGS (\d+)', header).group(1)) == _hip.CALL_MAX_ARGS
  for name in _hip.FN_IDS:
    assert len(_hip.PROTOS[name]) <= _hip.CALL_MAX_ARGS, name


class _FakeAcc:
  def __init__(self):
    self.blocks, self._host_const, self._recorder = [], {}, None


def test_recorder_refuses_what_is_not_steady_state():
  _hip.load_library()
  a = np.zeros(8, np.float32)
  da = xr.DataArray(a, dims=('x',))
  base = a.__array_interface__['data'][0]

  def build(log, **kw):
    rec = replay.ChunkRecorder({(0, 'p', 'v'): da}, _FakeAcc())
    da.__dict__['_wbx_dev'] = {0: type('D', (), {'ptr': base, 'nbytes': 32, 'fence': None})()}
    rec.log = list(log)
    for k, v in kw.items():
      setattr(rec, k, v)
    replay.reset_stats()
    return rec.finish(), list(replay.STATS['refusals'])

  ok_call = ('wbx_acc_add', (C.c_void_p(0), C.c_void_p(base + 8), C.c_void_p(base), 2, 0))
  det = ('wbx_memset', (C.c_void_p(0), C.c_void_p(base + 4), 0, 8))
  rec, why = build([det, ok_call])
  assert rec is not None and why == []
  assert [t for t in rec.slot_tags] == [((0, 'p', 'v'), 'data', 0)] and rec.nrelocs == 3
  assert sorted((r.call, r.arg, r.offset) for r in list(rec.relocs)[:3]) == [(0, 1, 4), (1, 1, 8), (1, 2, 0)]
  rec, why = build([('wbx_malloc', (None, 8, None)), det])  # an allocation stays with the record: nothing to replay, nothing to refuse
  assert rec is not None and len(rec.calls) == 1
  for log, text in (([('wbx_free', (None, None)), det], 'not an enqueue-only entry point'),
                    ([('wbx_memcpy_h2d', (None, None, None, 8)), det], 'not an enqueue-only entry point'),
                    ([ok_call], 'no launch was seen'),
                    ([('wbx_memset', (C.c_void_p(0), C.c_void_p(0xdead0000), 0, 8)), det], 'does not own')):
    rec, why = build(log)
    assert rec is None and text in why[0], (why, text)
  rec, why = build([det], refusal='an accumulator slot was created in the chunk')
  assert rec is None and 'accumulator slot' in why[0]
  # queries are left out; a pointer into memory the record keeps alive is fine
  table = np.zeros(4, np.int64)
  rec, why = build([('wbx_s1_partial_len', (None, 3, None)), det,
                    ('wbx_memset', (C.c_void_p(0), C.c_void_p(table.__array_interface__['data'][0] + 8), 0, 8))], kept=[{'t': [table]}])
  assert rec is not None and len(rec.calls) == 2
  da.__dict__.pop('_wbx_dev')


def test_chunk_signature_follows_coordinate_content_not_addresses():
  """two chunks whose (large) coordinates have the same shape and sit at the same ADDRESS but hold other values --
  station coordinates over `index`, a buffer the loader rewrites in place -- must not share a chunk record: Regions / area
  weights / atom tables are built from exactly these numbers."""
  n = 6000  # (> 4096 elements: the size from which round 5 hashed the data pointer)
  lat = np.linspace(-60, 60, n)
  payload = np.zeros(n, np.float32)
  a = xr.DataArray(payload, dims=('index',), coords={'latitude': ('index', lat), 'index': np.arange(n)})
  sig_a = pipeline._array_signature(a)
  lat[:] = np.linspace(-30, 30, n)  # rewritten in place: same object, same address
  b = xr.DataArray(payload, dims=('index',), coords={'latitude': ('index', lat), 'index': np.arange(n)})
  sig_b = pipeline._array_signature(b)
  assert sig_a != sig_b
  lat[:] = np.linspace(-60, 60, n)
  c = xr.DataArray(payload, dims=('index',), coords={'latitude': ('index', lat.copy()), 'index': np.arange(n)})
  assert pipeline._array_signature(c) == sig_a  # equal content in fresh arrays: the same signature (records carry over)
  # a frozen array is hashed once per object
  frozen = np.linspace(0, 1, n)
  frozen.flags.writeable = False
  t1 = pipeline._coord_token(frozen)
  assert pipeline._token_memo[id(frozen)][0] is frozen and pipeline._coord_token(frozen) is t1
  torch = pytest.importorskip('torch')
  big = torch.zeros(5000)
  assert pipeline._coord_token(big) != pipeline._coord_token(big)  # large tensors: never replayed
  small = torch.arange(10.0)
  assert pipeline._coord_token(small) == pipeline._coord_token(small.clone())


# ------------------------------------------------------------------------------------------------------------------- GPU
def _torch():
  import torch
  return torch


def _det_job(nchunks, nlat=73, nlon=144, nlead=3, nlev=2, change_at=None, layout='lon_fastest'):
  torch = _torch()
  g = torch.Generator(device='cuda')
  g.manual_seed(5)
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  shp = tuple({'latitude': nlat, 'longitude': nlon}[d] for d in sp)
  lead = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(nchunks) * np.timedelta64(24, 'h')
  level = np.arange(nlev)
  dims = ('init_time', 'lead_time', 'level') + sp
  pool = [(torch.randn((1, nlead, nlev) + shp, generator=g, device='cuda') + 280, torch.randn((1, nlead, nlev) + shp, generator=g, device='cuda') + 280)
          for _ in range(3)]
  hole = torch.rand((1, nlead, nlev) + shp, generator=g, device='cuda') < 0.05
  cdims = ('dayofyear', 'hour', 'level') + sp
  clim = xr.Dataset({'z': xr.DataArray(torch.randn((nchunks + 4, 4, nlev) + shp, generator=g, device='cuda') * 3 + 280, dims=cdims, coords={
      'dayofyear': np.arange(1, nchunks + 5), 'hour': np.array([0, 6, 12, 18]), 'level': level, 'latitude': lat, 'longitude': lon})})
  land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]) > 0.35
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  index = {int(t.astype('int64')): i for i, t in enumerate(inits)}

  def load(init_chunk, lead_chunk):
    i = index[int(init_chunk[0].astype('int64'))]
    p_t, t_t = pool[i % 3]
    nl = nlead
    if change_at is not None and i >= change_at:  # another chunk shape from here on: the record no longer applies
      nl = nlead - 1
      p_t, t_t = p_t[:, :nl].contiguous(), t_t[:, :nl].contiguous()
    cs = {'init_time': init_chunk, 'lead_time': lead[:nl], 'level': level, 'latitude': lat, 'longitude': lon}
    t = xr.DataArray(t_t, dims=dims, coords=cs)
    t = t.assign_coords(mask=xr.DataArray(~hole[:, :nl], dims=dims))
    return {'z': xr.DataArray(p_t, dims=dims, coords=cs)}, {'z': t}

  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias(), 'acc': deterministic.ACC(clim),
             'activity': deterministic.PredictionActivity(clim)}
  aggs = {'regions': aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                                            bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True),
          'plain': aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])}
  times = time_chunks.TimeChunks(inits, lead if change_at is None else lead, init_time_chunk_size=1)
  return times, load, metrics, aggs


def _ens_job(nchunks, with_mask, layout='lon_fastest', nlat=73, nlon=144, nlead=2, m=9, skipna=False):
  torch = _torch()
  g = torch.Generator(device='cuda')
  g.manual_seed(6)
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  shp = tuple({'latitude': nlat, 'longitude': nlon}[d] for d in sp)
  lead = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(nchunks) * np.timedelta64(24, 'h')
  pd, td = ('init_time', 'lead_time', 'number') + sp, ('init_time', 'lead_time') + sp
  pool = []
  for _ in range(2):
    t_t = torch.randn((1, nlead) + shp, generator=g, device='cuda') + 280
    p_t = t_t[:, :, None] + torch.randn((1, nlead, m) + shp, generator=g, device='cuda')
    if with_mask == 'nan' or skipna:
      t_t = t_t.clone()
      t_t[torch.rand(t_t.shape, generator=g, device='cuda') < 0.1] = float('nan')
    pool.append((p_t, t_t))
  valid = torch.rand(shp, generator=g, device='cuda') > 0.2
  land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]) > 0.35
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  index = {int(t.astype('int64')): i for i, t in enumerate(inits)}

  def load(init_chunk, lead_chunk):
    i = index[int(init_chunk[0].astype('int64'))]
    p_t, t_t = pool[i % 2]
    cs = {'init_time': init_chunk, 'lead_time': lead, 'latitude': lat, 'longitude': lon}
    t = xr.DataArray(t_t, dims=td, coords=cs)
    if with_mask == 'nan':
      from weatherbenchx_amd import data as wdata
      t = wdata.add_nan_mask_to_data({'v': t})['v']
    elif with_mask:
      t = t.assign_coords(mask=xr.DataArray(valid, dims=sp))
    return {'v': xr.DataArray(p_t, dims=pd, coords=cs)}, {'v': t}

  metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio(),
             'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')])}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True, skipna=skipna)
  return time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1), load, metrics, agg


def _states_equal(a, b):
  assert set(a) == set(b)
  for name in a:
    for tree_a, tree_b in ((a[name].sum_weighted_statistics, b[name].sum_weighted_statistics), (a[name].sum_weights, b[name].sum_weights)):
      assert set(tree_a) == set(tree_b)
      for stat in tree_a:
        for var in tree_a[stat]:
          x, y = tree_a[stat][var], tree_b[stat][var]
          assert x.dims == y.dims
          np.testing.assert_array_equal(np.asarray(x.values), np.asarray(y.values), err_msg=f'{name} {stat} {var}')  # bit for bit


def _replays(n):
  """Chunks of a loop alternate between two signatures (which launch stream an ensemble launch takes, engine.ALTERNATE_CHUNKS):
  of each kind the first builds (plans, tables, accumulator slots), the second is recorded, the rest are replayed."""
  return max(0, (n + 1) // 2 - 2) + max(0, n // 2 - 2)


def _on_off(job, monkeypatch, nchunks, expect_replays):
  engine.clear_caches()
  replay.reset_stats()
  monkeypatch.setattr(replay, 'ENABLED', True)
  times, load, metrics, aggs = job()
  on = pipeline.evaluate_chunks(times, load, metrics, aggs)
  stats = dict(replay.STATS)
  engine.clear_caches()
  monkeypatch.setattr(replay, 'ENABLED', False)
  times, load, metrics, aggs = job()
  off = pipeline.evaluate_chunks(times, load, metrics, aggs)
  _states_equal(on, off)
  # (a recording that met an allocation -- a pool filling up behind an earlier record's pinned blocks -- is tried again on the
  #  next chunk of its kind: up to two replays fewer)
  assert stats['recorded'] >= 1 and expect_replays - 2 <= stats['replayed'] <= expect_replays and stats['replayed'] >= 1, stats
  return on, stats


@pytest.mark.gpu
@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_deterministic_loop_with_climatology_bins_and_mask_replays(monkeypatch, layout):
  """RMSE / MAE / bias / ACC / activity, two aggregators (34-bin style regions with a per-point mask; plain area mean), a
  climatology whose gather table follows every chunk's time labels: chunk 1 builds, chunk 2 is recorded, chunks 3.. are replayed;
  the accumulators equal the ordinary path's bit for bit."""
  n = 9
  _on_off(lambda: _det_job(n, layout=layout), monkeypatch, n, _replays(n))


@pytest.mark.gpu
@pytest.mark.parametrize('with_mask', [False, True, 'nan'])
@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_ensemble_loop_replays(monkeypatch, with_mask, layout):
  n = 8
  _on_off(lambda: _ens_job(n, with_mask, layout), monkeypatch, n, _replays(n))


@pytest.mark.gpu
def test_ensemble_skipna_loop_replays(monkeypatch):
  n = 7
  _on_off(lambda: _ens_job(n, True, skipna=True), monkeypatch, n, _replays(n))


@pytest.mark.gpu
def test_changed_chunk_shape_takes_the_ordinary_path_again(monkeypatch):
  """Chunks 0-7 have three lead times, chunks 8-15 two (other arrays, other plans): the first records do not apply to them; they
  build, record and replay their own."""
  n = 16
  # (lead_time survives nowhere: both shapes add into the same per-(level, region) accumulators ... of different lead counts --
  #  reduce lead_time too so that the sums are comparable)
  def job():
    times, load, metrics, aggs = _det_job(n, change_at=8)
    aggs = {'plain': aggregation.Aggregator(reduce_dims=['init_time', 'lead_time', 'latitude', 'longitude'],
                                            weigh_by=[weighting.GridAreaWeighting()])}
    metrics = {k: metrics[k] for k in ('rmse', 'mae', 'bias')}
    return times, load, metrics, aggs
  _, stats = _on_off(job, monkeypatch, n, 2 * _replays(8))
  assert stats['recorded'] >= 3


@pytest.mark.gpu
def test_passes_with_spectra_fused_into_the_deterministic_sweep_replay(monkeypatch):
  """configs[4] in miniature on 1440-point rows: deterministic suite + zonal spectra of p and t (ONE fused sweep) + an ensemble
  suite from another loader, as one job."""
  torch = _torch()
  n, nlat, nlon, nlead, nlev, m = 10, 31, 1440, 2, 2, 5
  g = torch.Generator(device='cuda')
  g.manual_seed(8)
  lat, lon = np.linspace(-75, 75, nlat), np.linspace(0, 360, nlon, endpoint=False)
  lead = (np.arange(nlead) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(n) * np.timedelta64(24, 'h')
  level = np.arange(nlev)
  zd = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  pool = [tuple(torch.randn((1, nlead, nlev, nlat, nlon), generator=g, device='cuda') + 280 for _ in range(2)) for _ in range(2)]
  epool = []
  for _ in range(2):
    t_t = torch.randn((1, nlead, nlat, nlon), generator=g, device='cuda') + 280
    epool.append((t_t[:, :, None] + torch.randn((1, nlead, m, nlat, nlon), generator=g, device='cuda'), t_t))
  clim = xr.Dataset({'z': xr.DataArray(torch.randn((n + 3, 4, nlev, nlat, nlon), generator=g, device='cuda') + 280,
                                       dims=('dayofyear', 'hour', 'level', 'latitude', 'longitude'),
                                       coords={'dayofyear': np.arange(1, n + 4), 'hour': np.array([0, 6, 12, 18]), 'level': level,
                                               'latitude': lat, 'longitude': lon})})
  index = {int(t.astype('int64')): i for i, t in enumerate(inits)}

  def load_det(ic, lc):
    i = index[int(ic[0].astype('int64'))]
    cs = {'init_time': ic, 'lead_time': lead, 'level': level, 'latitude': lat, 'longitude': lon}
    return {'z': xr.DataArray(pool[i % 2][0], dims=zd, coords=cs)}, {'z': xr.DataArray(pool[i % 2][1], dims=zd, coords=cs)}

  def load_ens(ic, lc):
    i = index[int(ic[0].astype('int64'))]
    cs = {'init_time': ic, 'lead_time': lead, 'latitude': lat, 'longitude': lon}
    return ({'t2m': xr.DataArray(epool[i % 2][0], dims=('init_time', 'lead_time', 'number', 'latitude', 'longitude'), coords=cs)},
            {'t2m': xr.DataArray(epool[i % 2][1], dims=('init_time', 'lead_time', 'latitude', 'longitude'), coords=cs)})

  def run():
    det = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias(), 'acc': deterministic.ACC(clim)}
    spec = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalPowerSpectrum('targets')}
    ens = {'crps': probabilistic.CRPSEnsemble(use_sort=True)}
    area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
    zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
    passes = [('deterministic', load_det, det, area), ('spectra', load_det, spec, zonal), ('ensemble', load_ens, ens, area)]
    times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
    out = pipeline.evaluate_passes(times, passes)
    return {name: out[name][None] for name in out}

  engine.clear_caches()
  replay.reset_stats()
  monkeypatch.setattr(replay, 'ENABLED', True)
  on = run()
  stats = dict(replay.STATS)
  engine.clear_caches()
  monkeypatch.setattr(replay, 'ENABLED', False)
  off = run()
  assert _replays(n) - 2 <= stats['replayed'] <= _replays(n) and stats['replayed'] >= 1, stats
  # the deterministic lanes and the ensemble sums are order-fixed: bit for bit; the spectra's accumulate path adds with fp64
  # atomics (the one order-dependent sum of the library): equal to rounding
  for name in ('deterministic', 'ensemble'):
    _states_equal({name: on[name]}, {name: off[name]})
  for stat in on['spectra'].sum_weighted_statistics:
    a = np.asarray(on['spectra'].sum_weighted_statistics[stat]['z'].values)
    b = np.asarray(off['spectra'].sum_weighted_statistics[stat]['z'].values)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12 * np.abs(b).max())


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['det', 'ens', 'ens_mask', 'ens_skipna'])
def test_launches_that_add_into_the_accumulator_themselves(monkeypatch, kind):
  """WBX_BINNED_ACCUMULATE (ABI 12): from the second chunk on wbx_det_binned / wbx_ens_binned get the result's accumulator slot as
  `out` and add into it in the kernel that forms the sums -- no scratch round trip and no wbx_acc_add launch.  The accumulators of
  a loop are bit for bit those of the scratch + wbx_acc_add path (the same fp64 addition, made by another kernel), with and
  without chunk records, and the recorded chunks hold no wbx_acc_add for those results."""
  n = 8
  job = {'det': lambda: _det_job(n), 'ens': lambda: _ens_job(n, False), 'ens_mask': lambda: _ens_job(n, True),
         'ens_skipna': lambda: _ens_job(n, True, skipna=True)}[kind]
  results = {}
  for fused in (True, False):
    for rep in (True, False):
      engine.clear_caches()
      replay.reset_stats()
      monkeypatch.setattr(engine, 'FUSED_ACC_ADD', fused)
      monkeypatch.setattr(replay, 'ENABLED', rep)
      seen = []
      real = replay.ChunkRecord.__init__

      def spy(self, calls, *a, _real=real, _seen=seen, **k):
        _seen.append([c[0] for c in calls])
        _real(self, calls, *a, **k)
      monkeypatch.setattr(replay.ChunkRecord, '__init__', spy)
      times, load, metrics, aggs = job()
      results[fused, rep] = pipeline.evaluate_chunks(times, load, metrics, aggs)
      monkeypatch.setattr(replay.ChunkRecord, '__init__', real)
      if rep:
        assert seen, 'no chunk was recorded'
        binned = sum(name in ('wbx_det_binned', 'wbx_ens_binned') for name in seen[0])
        adds = sum(name == 'wbx_acc_add' for name in seen[0])
        assert binned >= 1
        results[fused, 'adds'] = (binned, adds)
  assert results[True, 'adds'][1] <= results[False, 'adds'][1] - results[True, 'adds'][0], results  # one add less per binned launch
  for key in ((True, False), (False, True), (False, False)):
    _states_equal(results[True, True], results[key])
