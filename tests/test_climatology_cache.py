"""Climatology slab cache (weatherbenchx_amd/climatology_cache.py): a host-resident [dayofyear, hour, level,
lat, lon] climatology behind a device pool of K slabs -- what `climatology.sel(dayofyear, hour).compute()` over a lazily
backed dataset is in the reference (metrics/base.py:396-403, data_loaders/xarray_loaders.py:266-316).

Both backends: the slot table (LRU, protection of slabs a chunk in the making names, invalidation), ACC / activity through
`evaluate_chunks` against the float64 oracle with a 366-day x 4-hour climatology and a pool SMALLER than the job's distinct
slabs, uploads == distinct slabs for a job that walks the calendar once, memory maps picked up by themselves.  `-m gpu`: the same
job with chunk records on (the gather table of a replayed chunk is built from pool slots), bit-identical to records off."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import wbx_oracle as O  # noqa: E402

from weatherbenchx_amd import aggregation  # noqa: E402
from weatherbenchx_amd import climatology_cache  # noqa: E402
from weatherbenchx_amd import pipeline  # noqa: E402
from weatherbenchx_amd import replay  # noqa: E402
from weatherbenchx_amd import time_chunks  # noqa: E402
from weatherbenchx_amd import weighting  # noqa: E402
from weatherbenchx_amd import xarray_lite as xr  # noqa: E402
from weatherbenchx_amd.metrics import base as metrics_base  # noqa: E402
from weatherbenchx_amd.metrics import deterministic  # noqa: E402

NLAT, NLON, NLEV = 9, 16, 2
LAT = np.linspace(-80, 80, NLAT)
LON = np.linspace(0, 360, NLON, endpoint=False)
LEVEL = np.array([500, 850])
HOURS = np.array([0, 6, 12, 18])
CDIMS = ('dayofyear', 'hour', 'level', 'latitude', 'longitude')
ZDIMS = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')


def _climatology(ndoy=366, dtype=np.float32, seed=5, order=CDIMS):
  rng = np.random.default_rng(seed)
  full = (280 + 10 * rng.standard_normal((ndoy, 4, NLEV, NLAT, NLON))).astype(dtype)
  coords = {'dayofyear': np.arange(1, ndoy + 1), 'hour': HOURS, 'level': LEVEL, 'latitude': LAT, 'longitude': LON}
  da = xr.DataArray(full, dims=CDIMS, coords=coords)
  if tuple(order) != CDIMS:
    da = da.transpose(*order)
  return full, da


def _job(ninit, nlead, lead_hours=6, init_step_hours=24, start='2020-01-01T00', seed=11):
  rng = np.random.default_rng(seed)
  lead = (np.arange(nlead) * lead_hours).astype('timedelta64[h]').astype('timedelta64[ns]')
  inits = np.datetime64(start, 'ns') + np.arange(ninit) * np.timedelta64(init_step_hours, 'h')
  p = (280 + 10 * rng.standard_normal((ninit, nlead, NLEV, NLAT, NLON))).astype(np.float32)
  t = (280 + 10 * rng.standard_normal((ninit, nlead, NLEV, NLAT, NLON))).astype(np.float32)
  index = {int(x.astype('int64')): i for i, x in enumerate(inits)}

  def load(ic, lc):
    ii = [index[int(x.astype('int64'))] for x in np.asarray(ic, 'datetime64[ns]')]
    cs = {'init_time': inits[ii], 'lead_time': lead, 'level': LEVEL, 'latitude': LAT, 'longitude': LON}
    return ({'z': xr.DataArray(p[ii].copy(), dims=ZDIMS, coords=cs)}, {'z': xr.DataArray(t[ii].copy(), dims=ZDIMS, coords=cs)})
  return inits, lead, p, t, load


def _oracle_acc(p, t, full, inits, lead):
  """ACC / activity / RMSE per (lead, level), area-weighted over (init, lat, lon), through the float64 oracle: the alignment
  (metrics/base.py:382-403), the anomaly statistics (deterministic.py:222-259) and the weighted reduction (aggregation.py:297-366)."""
  valid = inits[:, None] + lead[None, :]
  c, cdims = O.align_climatology(full, CDIMS, valid, ('init_time', 'lead_time'))
  assert cdims == ZDIMS
  w = (O.grid_area_weights(LAT), ('latitude',))
  reduce = ['init_time', 'latitude', 'longitude']

  def mean(stat):
    sws, sw, od = O.aggregate(stat, ZDIMS, reduce, weights=[w])
    assert tuple(od) == ('lead_time', 'level')
    return sws / sw
  cov, spa, sta = mean(O.anomaly_covariance(p, t, c)), mean(O.squared_prediction_anomaly(p, c)), mean(O.squared_target_anomaly(t, c))
  return {'acc': O.acc(cov, spa, sta), 'activity': np.sqrt(spa), 'rmse': O.rmse(mean(O.squared_error(p, t)))}


def _metrics(clim_ds):
  return {'acc': deterministic.ACC(clim_ds), 'activity': deterministic.PredictionActivity(clim_ds), 'rmse': deterministic.RMSE()}


def _area():
  return aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])


def _check(values, want, rtol=1e-6):
  for name, key in (('acc', 'acc.z'), ('activity', 'activity.z'), ('rmse', 'rmse.z')):
    got = values[key].transpose('lead_time', 'level').values
    np.testing.assert_allclose(got, want[name], rtol=rtol, atol=0, err_msg=name)


def test_slot_table_lru_and_protection(backend):
  full, da = _climatology(ndoy=6)
  cache = climatology_cache.SlabCache(da, slots=5)
  over = ('init_time', 'lead_time')

  def ask(days, hours, prefetch=False, launch=True):
    pos = {'dayofyear': np.asarray(days)[None, :], 'hour': np.asarray(hours)[None, :]}
    ref = cache.ref(over, pos, prefetch=prefetch)
    if launch:
      ref.activate()  # (what lazy.gather_from_ref does in front of a launch)
    return ref, ref.positions[climatology_cache.SLAB_DIM].reshape(-1)

  with climatology_cache.chunk_loop():
    ref, slots = ask([0, 0, 1, 1], [0, 2, 0, 2], launch=False)
    assert slots.tolist() == [-1] * 4 and cache.stats['uploads'] == 0  # nothing moves before a launch asks (or a prefetch)
    ref.activate()
    assert sorted(slots.tolist()) == [0, 1, 2, 3] and cache.stats['uploads'] == 4 and cache.stats['hits'] == 0
    assert ref.source.dims == (climatology_cache.SLAB_DIM, 'level', 'latitude', 'longitude')
    assert ref.aligned_dims() == over + ('level', 'latitude', 'longitude')
    # what a user-defined statistic is handed: the HOST gather (never the pool)
    np.testing.assert_array_equal(ref.aligned_view().values[0], full[[0, 0, 1, 1], [0, 2, 0, 2]])
    _, slots2 = ask([0, 1], [0, 0])
    assert cache.stats['uploads'] == 4 and cache.stats['hits'] == 2 and slots2.tolist() == [slots[0], slots[2]]
    # the four slabs are named by launches of the chunk in the making: a fifth fits, a sixth does not ...
    ask([2], [1])
    with pytest.raises(ValueError, match='slots cannot hold'):
      ask([3], [1])
    # ... and one asked for AHEAD is simply left out
    ahead, _ = ask([3], [1], prefetch=True, launch=False)
    assert cache.stats['uploads'] == 5 and (3, 1) not in cache.slot_of
    cache.chunk_enqueued()
    ahead.activate()
    # LRU: (0, 2) had been named before (0, 0) / (1, 0) were named again
    assert cache.stats['evictions'] == 1 and (0, 2) not in cache.slot_of and (0, 0) in cache.slot_of and (1, 0) in cache.slot_of
    # a table is written when its launch is made: `ref` (made five requests ago) finds (0, 2) gone and brings it back
    cache.chunk_enqueued()
    ref.activate()
    assert cache.stats['uploads'] == 7 and sorted(cache.slot_of[k] for k in [(0, 0), (0, 2), (1, 0), (1, 2)]) == sorted(slots.tolist())
    with pytest.raises(ValueError, match='one request names'):
      ask([0, 1, 2, 3, 4, 5], [0, 0, 0, 0, 0, 0])
  # outside a chunk loop a launch follows its table at once: earlier requests protect nothing
  for d in range(6):
    for h in range(4):
      ask([d], [h])
  assert cache.stats['evictions'] >= 19


def test_an_edited_climatology_gets_a_fresh_pool(backend):
  """`da[...] = x` drops what the engine cached on the object -- the pool with its stale slabs too; the wish for a pool stays."""
  full, da = _climatology(ndoy=4)
  clim = climatology_cache.cached(xr.Dataset({'z': da}), slots=3)
  inits, lead, p, t, load = _job(2, 2, lead_hours=6)
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  values = pipeline.evaluate_chunks(times, load, _metrics(clim), _area())[None].metric_values(_metrics(clim))
  _check(values, _oracle_acc(p, t, full, inits, lead))
  da = clim['z']  # (the Dataset holds its own DataArray object over the same payload)
  first = climatology_cache.cache_for(da)
  assert first is not None and first.stats['uploads'] >= 4
  before = full.copy()
  da[0, 0] = da.values[0, 0] + 5.0  # (`full` is the payload itself)
  full2 = np.asarray(da.values)
  assert np.all(full2[0, 0] == before[0, 0] + 5.0)
  values = pipeline.evaluate_chunks(times, load, _metrics(clim), _area())[None].metric_values(_metrics(clim))
  _check(values, _oracle_acc(p, t, full2, inits, lead))
  second = climatology_cache.cache_for(da)
  assert second is not first and second.nslots == 3


@pytest.mark.parametrize('order', [CDIMS, ('hour', 'level', 'dayofyear', 'latitude', 'longitude')])
def test_acc_over_a_year_with_a_pool_smaller_than_the_job(backend, order):
  """366 x 4 slabs on the host, 40 daily inits x 5 six-hourly leads = 161 distinct slabs, 12 slots: every slab is uploaded
  ONCE (the job walks the calendar), 149 evictions, ACC / activity equal the float64 oracle at 1e-6; a strided source (the
  selected dims are not in front) goes through the staging copy."""
  replay.reset_stats()
  full, da = _climatology(order=order)
  clim = climatology_cache.cached(xr.Dataset({'z': da}), slots=12)
  inits, lead, p, t, load = _job(40, 5, start='2020-02-20T00')  # (crosses Feb 29)
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  states = pipeline.evaluate_chunks(times, load, _metrics(clim), _area())
  values = states[None].metric_values(_metrics(clim))
  _check(values, _oracle_acc(p, t, full, inits, lead))
  cache = climatology_cache.cache_for(xr.as_dataarray(clim['z']))
  assert cache.nslots == 12
  distinct = 40 * 4 + 1  # 00/06/12/18 of each init day; 00 of the next day is shared with the next init
  assert cache.stats['uploads'] == distinct, cache.stats
  assert cache.stats['evictions'] == distinct - 12
  assert cache.stats['prefetched'] >= distinct - 5  # all but the first chunk's slabs were asked for one chunk ahead
  assert cache.stats['upload_bytes'] == distinct * NLEV * NLAT * NLON * 4


def test_results_equal_the_resident_climatology(backend):
  full, da = _climatology(ndoy=20)
  inits, lead, p, t, load = _job(12, 4, lead_hours=12)
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=2)
  resident = xr.Dataset({'z': da})
  a = pipeline.evaluate_chunks(times, load, _metrics(resident), _area())[None]
  _, da2 = _climatology(ndoy=20)
  pooled = climatology_cache.cached(xr.Dataset({'z': da2}), slots=9)
  b = pipeline.evaluate_chunks(times, load, _metrics(pooled), _area())[None]
  for kind in ('sum_weighted_statistics', 'sum_weights'):
    ta, tb = getattr(a, kind), getattr(b, kind)
    for stat in ta:
      np.testing.assert_array_equal(np.asarray(ta[stat]['z'].values), np.asarray(tb[stat]['z'].values), err_msg=f'{kind} {stat}')
  assert climatology_cache.cache_for(pooled['z']).stats['evictions'] > 0


def test_a_latitude_fastest_climatology_is_transposed_on_its_way_into_the_pool(backend):
  """Archive order [.., longitude, latitude] for fields AND climatology, `device_layout='lon_fastest'` for both: the pool holds
  [level, latitude, longitude] slabs (the staging copy is wbx_host_transpose), the statistics see one layout."""
  full, _ = _climatology(ndoy=12)
  coords = {'dayofyear': np.arange(1, 13), 'hour': HOURS, 'level': LEVEL, 'latitude': LAT, 'longitude': LON}
  lat_fast = xr.DataArray(np.ascontiguousarray(np.swapaxes(full, -1, -2)), dims=CDIMS[:3] + ('longitude', 'latitude'), coords=coords)
  clim = climatology_cache.cached(xr.Dataset({'z': lat_fast}), slots=8, device_layout='lon_fastest')
  inits, lead, p, t, load = _job(8, 3, lead_hours=12)
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  values = pipeline.evaluate_chunks(times, load, _metrics(clim), _area())[None].metric_values(_metrics(clim))
  _check(values, _oracle_acc(p, t, full, inits, lead))
  cache = climatology_cache.cache_for(clim['z'])
  assert cache.slab_dims == ('level', 'latitude', 'longitude') and cache.stats['evictions'] > 0


def test_memory_maps_get_a_pool_by_themselves(backend, tmp_path, monkeypatch):
  full, _ = _climatology(ndoy=10)
  path = str(tmp_path / 'clim.npy')
  np.save(path, full)
  mm = np.load(path, mmap_mode='r')
  coords = {'dayofyear': np.arange(1, 11), 'hour': HOURS, 'level': LEVEL, 'latitude': LAT, 'longitude': LON}
  monkeypatch.setattr(climatology_cache, 'AUTO_POOL_BYTES', 7 * NLEV * NLAT * NLON * 4)
  da = xr.DataArray(mm, dims=CDIMS, coords=coords)
  clim = xr.Dataset({'z': da})
  inits, lead, p, t, load = _job(6, 3, lead_hours=12)
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  values = pipeline.evaluate_chunks(times, load, _metrics(clim), _area())[None].metric_values(_metrics(clim))
  _check(values, _oracle_acc(p, t, full, inits, lead))
  cache = climatology_cache.cache_for(xr.as_dataarray(clim['z']))
  assert cache is not None and cache.nslots == 7 and cache.stats['uploads'] == 6 * 2 + 1
  # a small in-memory climatology stays on the one-upload path
  _, small = _climatology(ndoy=3)
  assert climatology_cache.cache_for(small) is None


def test_single_chunk_api_and_dtype_mismatch(backend):
  """Outside a chunk loop (aggregation.compute_metric_values_for_single_chunk): nothing stamps the slots, an eviction waits for
  whatever has been enqueued.  A float64 field against a float32 pool is refused with the way out."""
  full, da = _climatology(ndoy=8)
  clim = climatology_cache.cached(xr.Dataset({'z': da}), slots=4)
  inits, lead, p, t, load = _job(6, 2, lead_hours=24)
  for i in range(6):
    pred, targ = load(inits[i:i + 1], None)
    values = aggregation.compute_metric_values_for_single_chunk(_metrics(clim), _area(), pred, targ)
    _check(values, _oracle_acc(p[i:i + 1], t[i:i + 1], full, inits[i:i + 1], lead))
    del pred, targ, values
  assert climatology_cache.cache_for(clim['z']).stats['evictions'] >= 3
  pred, targ = load(inits[:1], None)
  pred = {'z': pred['z'].astype(np.float64)}
  with pytest.raises(ValueError, match='cast the climatology'):
    aggregation.compute_metric_values_for_single_chunk({'acc': deterministic.ACC(clim)}, _area(), pred, targ)


def test_user_statistic_gets_the_host_gather(backend):
  full, da = _climatology(ndoy=5)
  clim = climatology_cache.cached(xr.Dataset({'z': da}), slots=3)

  class MeanAnomaly(metrics_base.PerVariableStatisticWithClimatology):
    @property
    def unique_name(self):
      return 'MeanAnomaly'

    def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
      return predictions - aligned_climatology.aligned_view()

  inits, lead, p, t, load = _job(2, 2, lead_hours=6)
  pred, targ = load(inits[:1], None)
  out = MeanAnomaly(clim).compute(pred, targ)['z']
  want = p[:1] - full[[0, 0], [0, 1]][None]
  np.testing.assert_allclose(out.transpose(*ZDIMS).values, want, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_replayed_chunks_address_the_pool(layout, monkeypatch):
  """Chunk records on: chunk 1 builds, chunk 2 is recorded, chunks 3.. are ONE library call each whose gather table holds pool
  slots; evictions happen between replays.  Bit-identical to records off; both equal the oracle."""
  from weatherbenchx_amd import _hip
  from weatherbenchx_amd import engine
  if not _hip.is_available():
    pytest.fail('gpu test selected but libwbx_hip.so / a HIP device is not available')
  import torch
  full, da = _climatology()
  nlat, nlon = (NLAT, NLON)
  inits, lead, p, t, _ = _job(30, 5, start='2020-12-20T00')  # (crosses the end of the leap year)
  perm = (0, 1, 2, 3, 4) if layout == 'lon_fastest' else (0, 1, 2, 4, 3)
  dims = tuple(ZDIMS[i] for i in perm)
  pd, td = torch.as_tensor(p).cuda().permute(*perm).contiguous(), torch.as_tensor(t).cuda().permute(*perm).contiguous()
  index = {int(x.astype('int64')): i for i, x in enumerate(inits)}

  def load(ic, lc):
    i = index[int(np.asarray(ic, 'datetime64[ns]')[0].astype('int64'))]
    cs = {'init_time': inits[i:i + 1], 'lead_time': lead, 'level': LEVEL, 'latitude': LAT, 'longitude': LON}
    return {'z': xr.DataArray(pd[i:i + 1], dims=dims, coords=cs)}, {'z': xr.DataArray(td[i:i + 1], dims=dims, coords=cs)}

  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  results = {}
  for on in (True, False):
    engine.clear_caches()
    monkeypatch.setattr(replay, 'ENABLED', on)
    replay.reset_stats()
    _, da_i = _climatology()
    clim = climatology_cache.cached(xr.Dataset({'z': da_i}), slots=11)
    state = pipeline.evaluate_chunks(times, load, _metrics(clim), _area())[None]
    cache = climatology_cache.cache_for(clim['z'])
    assert cache.stats['uploads'] == 30 * 4 + 1 and cache.stats['evictions'] == 30 * 4 + 1 - 11, cache.stats
    if on:
      assert replay.STATS['replayed'] >= 25, dict(replay.STATS)
    else:
      assert replay.STATS['replayed'] == 0
    results[on] = state
    _check(state.metric_values(_metrics(clim)), _oracle_acc(p, t, full, inits, lead))
  for kind in ('sum_weighted_statistics', 'sum_weights'):
    ta, tb = getattr(results[True], kind), getattr(results[False], kind)
    for stat in ta:
      np.testing.assert_array_equal(np.asarray(ta[stat]['z'].values), np.asarray(tb[stat]['z'].values), err_msg=f'{kind} {stat}')
  engine.clear_caches()


def test_time_indexed_climatology_and_two_variables(backend):
  """A climatology over a `time` axis (metrics/base.py:389-391 selects `time=valid_time`) and two variables with a pool each."""
  rng = np.random.default_rng(2)
  ntime = 40
  times_c = np.datetime64('2020-03-01T00', 'ns') + np.arange(ntime) * np.timedelta64(6, 'h')
  full = {v: (280 + 10 * rng.standard_normal((ntime, NLEV, NLAT, NLON))).astype(np.float32) for v in ('z', 'q')}
  coords = {'time': times_c, 'level': LEVEL, 'latitude': LAT, 'longitude': LON}
  clim = climatology_cache.cached(xr.Dataset({v: xr.DataArray(full[v], dims=('time', 'level', 'latitude', 'longitude'), coords=coords)
                                               for v in full}), slots=7)
  inits, lead, p, t, _ = _job(6, 4, lead_hours=6, start='2020-03-01T00')
  index = {int(x.astype('int64')): i for i, x in enumerate(inits)}

  def load(ic, lc):
    ii = [index[int(x.astype('int64'))] for x in np.asarray(ic, 'datetime64[ns]')]
    cs = {'init_time': inits[ii], 'lead_time': lead, 'level': LEVEL, 'latitude': LAT, 'longitude': LON}
    return ({v: xr.DataArray(p[ii] + k, dims=ZDIMS, coords=cs) for k, v in enumerate(full)},
            {v: xr.DataArray(t[ii] + k, dims=ZDIMS, coords=cs) for k, v in enumerate(full)})
  tc = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  values = pipeline.evaluate_chunks(tc, load, {'acc': deterministic.ACC(clim)}, _area())[None].metric_values({'acc': deterministic.ACC(clim)})
  valid = inits[:, None] + lead[None, :]
  pos = ((valid - times_c[0]) // np.timedelta64(6, 'h')).astype(int)
  w = O.grid_area_weights(LAT)[None, None, None, :, None]
  for k, v in enumerate(full):
    c = full[v][pos].astype(np.float64)
    pf, tf = p.astype(np.float64) + k, t.astype(np.float64) + k
    mean = lambda x: (x * w).sum(axis=(0, 3, 4)) / (np.ones_like(x) * w).sum(axis=(0, 3, 4))  # noqa: E731
    want = mean((pf - c) * (tf - c)) / np.sqrt(mean((pf - c) ** 2) * mean((tf - c) ** 2))
    np.testing.assert_allclose(values[f'acc.{v}'].transpose('lead_time', 'level').values, want, rtol=1e-6, err_msg=v)
    cache = climatology_cache.cache_for(clim[v])
    assert cache.sel_dims == ('time',) and cache.stats['evictions'] > 0
    assert cache.stats['uploads'] == len(np.unique(pos))


def test_the_wish_for_a_pool_travels_with_a_pickled_metric(backend):
  """Beam-style workers pickle their metrics (beam_pipeline.py:140-160): the pool and its thread stay behind, the wish
  (`cached(..., slots=)`) arrives and the worker builds its own pool on first use."""
  import pickle
  full, da = _climatology(ndoy=6)
  clim = climatology_cache.cached(xr.Dataset({'z': da}), slots=4)
  inits, lead, p, t, load = _job(3, 2, lead_hours=12)
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  metrics = _metrics(clim)
  pipeline.evaluate_chunks(times, load, metrics, _area())
  assert climatology_cache.cache_for(clim['z']).pool is not None
  there = pickle.loads(pickle.dumps(metrics))
  c2 = there['acc']._climatology['z']  # pylint: disable=protected-access
  assert '_wbx_slab_cache' not in c2.__dict__ and c2.__dict__['_slab_cache_config'][0] == 4
  values = pipeline.evaluate_chunks(times, load, there, _area())[None].metric_values(there)
  _check(values, _oracle_acc(p, t, full, inits, lead))
  cache = climatology_cache.cache_for(c2)
  assert cache is not climatology_cache.cache_for(clim['z']) and cache.nslots == 4 and cache.stats['uploads'] > 0
