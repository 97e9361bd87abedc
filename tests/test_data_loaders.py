"""In-memory loaders with the reference's names (weatherbenchx_amd/data_loaders/).  The cases are the ones of
weatherbenchX/data_loaders/xarray_loaders_test.py:24-160 on the same mock datasets (passed as `ds=`: there is no zarr here),
with the VALUES of every chunk checked against plain numpy indexing of the source arrays, plus the shared `load_chunk`
steps of data_loaders/base.py:119-170 and a chunked evaluation through them against the oracle."""
import numpy as np
import pytest

from oracle import wbx_oracle as O
from tests import mock_data
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import data_loaders
from weatherbenchx_amd import loaders
from weatherbenchx_amd import pipeline
from weatherbenchx_amd import time_chunks
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import deterministic

VARIABLES = ['geopotential', '2m_temperature']
DAY = np.timedelta64(1, 'D').astype('timedelta64[ns]')


def _inits(start, stop):
  return np.arange(start, stop, np.timedelta64(24, 'h'), dtype='datetime64[ns]')


def test_prediction_target_dimension_match():
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-20T00', random=True, seed=1)
  prediction = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True, seed=2)
  lt = data_loaders.TargetsFromXarray(ds=target, variables=VARIABLES)
  lp = data_loaders.PredictionsFromXarray(ds=prediction, variables=VARIABLES)
  init_times = _inits('2020-01-01T00', '2020-01-03T00')[::-1].copy()      # two inits, in the order ASKED for
  lead_times = np.arange(3, dtype='timedelta64[D]').astype('timedelta64[ns]')
  t, p = lt.load_chunk(init_times, lead_times), lp.load_chunk(init_times, lead_times)
  assert set(t) == set(p) == set(VARIABLES)
  for name in VARIABLES:
    assert t[name].sizes == p[name].sizes
    for d in t[name].dims:
      np.testing.assert_array_equal(t[name][d].values, p[name][d].values)
  # ecmwf renaming (xarray_loaders.py:36-40) and the frames
  tz, pz = t['geopotential'], p['geopotential']
  assert set(tz.dims) == {'init_time', 'lead_time', 'latitude', 'longitude', 'level'}
  np.testing.assert_array_equal(tz.coords['valid_time'].values, init_times[:, None] + lead_times[None, :])
  src_t, src_p = target['geopotential'], prediction['geopotential']      # dims (time, latitude, longitude, level) / lead first
  tt = tz.transpose('init_time', 'lead_time', 'latitude', 'longitude', 'level').values
  pp = pz.transpose('init_time', 'lead_time', 'latitude', 'longitude', 'level').values
  for a, i in enumerate((1, 0)):
    for b in range(3):
      np.testing.assert_array_equal(tt[a, b], src_t.values[i + b])
      np.testing.assert_array_equal(pp[a, b], src_p.values[b, i])


def test_prediction_lead_time_selection():
  prediction = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True, seed=3)
  lp = data_loaders.PredictionsFromXarray(ds=prediction, variables=['2m_temperature'])
  init_times = _inits('2020-01-01T00', '2020-01-02T00')
  assert lp.load_chunk(init_times)['2m_temperature'].sizes['lead_time'] == 11                  # None: every lead time
  sl = lp.load_chunk(init_times, slice(2 * DAY, 4 * DAY))['2m_temperature']                    # by label, both ends included
  np.testing.assert_array_equal(sl['lead_time'].values, np.array([2, 3, 4]) * DAY)
  with pytest.raises(KeyError):
    lp.load_chunk(init_times, np.array([36], dtype='timedelta64[h]'))
  with pytest.raises(KeyError):
    lp.load_chunk(np.array(['2031-01-01'], dtype='datetime64[ns]'), np.array([0], dtype='timedelta64[h]'))


def test_targets_without_lead_times_and_with_a_slice():
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-06T00', random=True, seed=4)
  lt = data_loaders.TargetsFromXarray(ds=target, variables=['2m_temperature'])
  times = _inits('2020-01-02T00', '2020-01-04T00')
  chunk = lt.load_chunk(times)['2m_temperature']                            # init times ARE the valid times (:271-274)
  assert chunk.dims[0] == 'valid_time'
  np.testing.assert_array_equal(chunk.values, target['2m_temperature'].values[1:3])
  with pytest.raises(ValueError, match='Lead time slice not supported'):
    lt.load_chunk(times, slice(None))


def test_climatology_loader():
  target = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-02T00').rename(
      time='init_time', prediction_timedelta='lead_time')
  climatology = target.isel(init_time=0, lead_time=0, drop=True).expand_dims(dayofyear=366, hour=4)
  init_times = _inits('2020-01-01T00', '2020-01-02T00')
  lead_times = np.arange(0, 3, 1, dtype='timedelta64[D]')
  loader = data_loaders.ClimatologyFromXarray(ds=climatology, variables=VARIABLES, climatology_time_coords=['dayofyear', 'hour'])
  chunk = loader.load_chunk(init_times, lead_times)
  assert set(chunk['geopotential'].dims) == {'init_time', 'lead_time', 'level', 'latitude', 'longitude'}
  assert set(chunk['2m_temperature'].dims) == {'init_time', 'lead_time', 'latitude', 'longitude'}


def test_climatology_values_follow_the_valid_time():
  rng = np.random.default_rng(6)
  doy, hour = np.arange(1, 367), np.array([0, 6, 12, 18])
  field = rng.normal(size=(366, 4, 5))
  clim = xr.Dataset({'t': xr.DataArray(field, dims=('dayofyear', 'hour', 'latitude'),
                                       coords={'dayofyear': doy, 'hour': hour, 'latitude': np.linspace(-60, 60, 5)})})
  loader = data_loaders.ClimatologyFromXarray(ds=clim)
  init_times = np.array(['2020-02-28T06', '2021-12-31T18'], dtype='datetime64[ns]')
  lead_times = np.array([0, 18, 48], dtype='timedelta64[h]')
  chunk = loader.load_chunk(init_times, lead_times)['t']
  assert chunk.dims == ('init_time', 'lead_time', 'latitude')
  #   2020-02-28T06 -> day 59 h 6 ; +18 h -> 2020-02-29T00 (day 60) ; +48 h -> 2020-03-01T06 (day 61: leap year)
  #   2021-12-31T18 -> day 365 h 18 ; +18 h -> 2022-01-01T12 (day 1) ; +48 h -> 2022-01-02T18 (day 2)
  expect = [[(59, 6), (60, 0), (61, 6)], [(365, 18), (1, 12), (2, 18)]]
  for a in range(2):
    for b in range(3):
      d, h = expect[a][b]
      np.testing.assert_array_equal(chunk.values[a, b], field[d - 1, h // 6])
  only_init = loader.load_chunk(init_times)['t']                            # no lead times: the init times are used (:322-327)
  assert only_init.dims == ('init_time', 'latitude')
  np.testing.assert_array_equal(only_init.values, np.stack([field[58, 1], field[364, 3]]))
  with pytest.raises(ValueError, match='slice'):
    loader.load_chunk(init_times, slice(None))


def test_persistence_loader():
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True, seed=7)
  init_times = _inits('2020-01-01T00', '2020-01-03T00')
  lead_times = np.arange(0, 3, 1, dtype='timedelta64[D]')
  loader = data_loaders.PersistenceFromXarray(ds=target, variables=VARIABLES)
  chunk = loader.load_chunk(init_times, lead_times)
  z = chunk['geopotential']
  assert set(z.dims) == {'init_time', 'lead_time', 'level', 'latitude', 'longitude'}
  np.testing.assert_array_equal(z['lead_time'].values, lead_times.astype('timedelta64[ns]'))
  zz = z.transpose('lead_time', 'init_time', 'latitude', 'longitude', 'level').values
  for b in range(3):
    np.testing.assert_array_equal(zz[b], target['geopotential'].values)     # the same analysis at every lead time
  for bad in (None, slice(None)):
    with pytest.raises(ValueError, match='Exact lead times'):
      loader.load_chunk(init_times, bad)


def test_probabilistic_climatology_loader():
  target = mock_data.mock_target_data(time_start='2015-01-01T00', time_stop='2021-01-01T00', variables_3d=[],
                                      variables_2d=['2m_temperature'], random=True, seed=8)
  init_times = _inits('2020-12-30T00', '2021-01-01T00')
  lead_times = np.arange(0, 3, 1, dtype='timedelta64[D]')
  loader = data_loaders.ProbabilisticClimatologyFromXarray(ds=target, start_year=2015, end_year=2019)
  chunk = loader.load_chunk(init_times, lead_times)['2m_temperature']
  assert set(chunk.dims) == {'number', 'init_time', 'lead_time', 'latitude', 'longitude'}
  assert chunk.sizes['number'] == 5
  np.testing.assert_array_equal(chunk['number'].values, np.arange(5))
  src = target['2m_temperature']
  times = src['valid_time' if 'valid_time' in src.coords else 'time'].values
  vals = chunk.transpose('number', 'init_time', 'lead_time', 'latitude', 'longitude').values
  # 2020-12-31 is day 366 of a leap year: in 2015 (365 days) that is 2016-01-01; in 2016 it is 2016-12-31
  valid = init_times[:, None] + lead_times[None, :].astype('timedelta64[ns]')
  for m, year in enumerate(range(2015, 2020)):
    for a in range(2):
      for b in range(3):
        v = valid[a, b]
        doy = int((v.astype('datetime64[D]') - v.astype('datetime64[Y]').astype('datetime64[D]')).astype(int)) + 1
        when = np.datetime64(str(year), 'D') + np.timedelta64(doy - 1, 'D')
        i = int(np.nonzero(times == when.astype('datetime64[ns]'))[0][0])
        np.testing.assert_array_equal(vals[m, a, b], src.values[i])
  assert np.datetime64('2016-01-01') in chunk.coords['valid_time'].values.astype('datetime64[D]')[0]  # member 2015, day 366


def test_constant_loader():
  constant = xr.Dataset({'2m_temperature': (('quantile',), [0.2, 0.4])}, coords={'quantile': [1, 2]})
  loader = data_loaders.ConstantLoader(constant_ds=constant)
  chunk = loader.load_chunk(_inits('2020-01-01T00', '2020-01-02T00'), np.arange(0, 3, 1, dtype='timedelta64[D]'))
  assert chunk is constant
  np.testing.assert_array_equal(chunk['2m_temperature'].values, [0.2, 0.4])


def test_constructor_errors_and_preparation_order():
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True, seed=9)
  with pytest.raises(ValueError, match='Only one of path or ds'):
    data_loaders.TargetsFromXarray(path='x.zarr', ds=target)
  with pytest.raises(ValueError, match='Either path or ds'):
    data_loaders.TargetsFromXarray()
  with pytest.raises(NotImplementedError, match='PredictionsFromFiles'):
    data_loaders.TargetsFromXarray(path='x.zarr')
  with pytest.raises(ValueError, match='rename_dimensions'):
    data_loaders.TargetsFromXarray(ds=target, rename_dimensions='cf').load_chunk(_inits('2020-01-01T00', '2020-01-02T00'))
  calls = []

  def pre(ds):
    calls.append('pre')
    return ds.rename({'latitude': 'lat', 'longitude': 'lon'})      # ... which the loader turns back (xarray_loaders.py:33-35)

  loader = data_loaders.TargetsFromXarray(ds=target, preprocessing_fn=pre, rename_variables={'2m_temperature': 't2m'},
                                          variables=['t2m'], sel_kwargs={'latitude': slice(-30, 30)})
  for _ in range(2):
    chunk = loader.load_chunk(_inits('2020-01-01T00', '2020-01-02T00'), np.array([0, 24], dtype='timedelta64[h]'))
  assert calls == ['pre']                                            # prepared once
  assert list(chunk) == ['t2m'] and 'latitude' in chunk['t2m'].dims
  np.testing.assert_array_equal(chunk['t2m']['latitude'].values, np.arange(-30, 31, 10.0))
  explicit = data_loaders.TargetsFromXarray(ds=target, rename_dimensions={'time': 'valid_time'}, variables=['geopotential'])
  assert explicit.load_chunk(_inits('2020-01-01T00', '2020-01-02T00'))['geopotential'].dims[0] == 'valid_time'


def test_shared_load_chunk_steps():
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-04T00', variables_3d=[], random=True,
                                      seed=10)
  target['2m_temperature'].values[1, 2, 3] = np.nan
  order = []

  class Interp:

    def interpolate(self, chunk, reference):
      order.append(('interp', reference))
      return {k: v.isel(longitude=slice(0, 4)) for k, v in chunk.items()}

  def process(chunk):
    order.append(('process', None))
    return {k: v + 1.0 for k, v in chunk.items()}

  loader = data_loaders.TargetsFromXarray(ds=target, interpolation=Interp(), process_chunk_fn=process, add_nan_mask=True,
                                          add_values_to_coords=True)
  init_times, lead_times = _inits('2020-01-01T00', '2020-01-03T00'), np.array([0, 24], dtype='timedelta64[h]')
  t = loader.load_chunk(init_times, lead_times, reference='REF')['2m_temperature']
  assert order == [('process', None), ('interp', 'REF')]                     # data_loaders/base.py:139-149
  assert t.sizes['longitude'] == 4
  np.testing.assert_array_equal(np.asarray(t.coords['mask'].values), ~np.isnan(t.values))
  assert not np.asarray(t.coords['mask'].values).all()
  np.testing.assert_array_equal(np.asarray(t.coords['values_as_coord'].values), t.values)
  src = target['2m_temperature'].values
  np.testing.assert_array_equal(t.transpose('init_time', 'lead_time', 'latitude', 'longitude').values[1, 1], src[2][:, :4] + 1.0)


def test_chunked_evaluation_through_the_in_memory_loaders(backend):
  del backend
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-12T00', random=True, seed=11)
  prediction = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-08T00', lead_stop_days=3,
                                              random=True, seed=12)
  target['geopotential'].values[4, 3, 2, 1] = np.nan
  lt = data_loaders.TargetsFromXarray(ds=target, variables=VARIABLES, add_nan_mask=True)
  lp = data_loaders.PredictionsFromXarray(ds=prediction, variables=VARIABLES)
  init_times = _inits('2020-01-01T00', '2020-01-08T00')
  lead_times = np.arange(4, dtype='timedelta64[D]').astype('timedelta64[ns]')
  tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=3, lead_time_chunk_size=2)
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               masked=True)
  got = pipeline.evaluate_chunks(tc, loaders.load_chunk_fn(lp, lt), metrics, agg, prefetch=0)[None].metric_values(metrics)
  # the whole arrays through the oracle: [init, lead, lat, lon, level]
  tz = target['geopotential'].values                                            # [time, lat, lon, level]
  pz = np.moveaxis(prediction['geopotential'].values, 0, 1)                     # [lead, time, ...] -> [time, lead, ...]
  tfull = tz[np.arange(7)[:, None] + np.arange(4)[None, :]]
  fdims = ('init_time', 'lead_time', 'latitude', 'longitude', 'level')
  w = (O.grid_area_weights(target['geopotential']['latitude'].values), ('latitude',))
  ok = ~np.isnan(tfull)
  for name, lane in (('rmse', O.squared_error(pz, tfull)), ('mae', O.absolute_error(pz, tfull))):
    sws, sw, od = O.aggregate(lane, fdims, ['init_time', 'latitude', 'longitude'], weights=[w], mask=ok, mask_dims=fdims)
    want = np.sqrt(sws / sw) if name == 'rmse' else sws / sw
    np.testing.assert_allclose(np.asarray(got[f'{name}.geopotential'].transpose(*od).values), want, rtol=1e-6, err_msg=name)


# ---- latency wrappers: the cases of weatherbenchX/data_loaders/latency_wrappers_test.py:24-325 -----------------------------

def _forecasts(time_start, time_stop, step_h, lead_stop_h, lead_step_h, seed=None, offset=0.0):
  """A forecast dataset [prediction_timedelta, time, latitude] with distinct values."""
  rng = np.random.default_rng(seed)
  times = np.arange(np.datetime64(time_start, 'ns'), np.datetime64(time_stop, 'ns'), np.timedelta64(step_h, 'h'))
  leads = (np.arange(0, lead_stop_h + 1, lead_step_h)).astype('timedelta64[h]').astype('timedelta64[ns]')
  vals = (rng.random((leads.size, times.size, 4)) if seed is not None else np.zeros((leads.size, times.size, 4))) + offset
  return xr.Dataset({'2m_temperature': xr.DataArray(vals, dims=('prediction_timedelta', 'time', 'latitude'),
                                                    coords={'prediction_timedelta': leads, 'time': times,
                                                            'latitude': np.linspace(-45, 45, 4)})})


def _h(values):
  return np.array(values, dtype='timedelta64[h]')


def _t(*stamps):
  return np.array(stamps, dtype='datetime64[ns]')


def test_latency_wrapper():
  prediction = _forecasts('2020-01-01T00', '2020-01-04T00', 12, 30, 6, seed=20)
  loader = data_loaders.PredictionsFromXarray(ds=prediction, variables=['2m_temperature'])
  init_times, lead_times, latency = _t('2020-01-02T00', '2020-01-02T06'), _h([6, 12]), np.timedelta64(6, 'h')
  available = [(_t('2020-01-01T12'), _h([18, 24])), (_t('2020-01-02T00'), _h([12, 18]))]
  explicit = data_loaders.ConstantLatencyWrapper(loader, latency=latency, nominal_init_times=prediction['time'].values)
  shorthand = data_loaders.XarrayConstantLatencyWrapper(loader, latency=latency)
  for wrapped in (explicit, shorthand):
    out = wrapped.load_chunk(init_times, lead_times)['2m_temperature']
    np.testing.assert_array_equal(out['init_time'].values, init_times)          # relabelled to what was ASKED for
    np.testing.assert_array_equal(out['lead_time'].values, lead_times.astype('timedelta64[ns]'))
    for i, (ai, al) in enumerate(available):
      want = loader.load_chunk(ai, al)['2m_temperature']
      np.testing.assert_array_equal(out.isel(init_time=[i]).values, want.values)
  assert explicit.get_available_init_time(np.datetime64('2020-01-02T05', 'ns')) == np.datetime64('2020-01-01T12', 'ns')
  assert explicit.get_available_init_time(np.datetime64('2020-01-01T05', 'ns')) is None
  with pytest.raises(ValueError, match='only valid with lead times'):
    explicit.load_chunk(init_times)
  with pytest.raises(ValueError, match='No available init time'):
    explicit.load_chunk(_t('2020-01-01T03'), lead_times)


def test_multiple_latency_wrappers():
  p0012 = _forecasts('2020-01-01T00', '2020-01-04T00', 12, 30, 6, seed=21)
  p0618 = _forecasts('2020-01-01T06', '2020-01-04T00', 12, 30, 6, seed=22)
  l0012 = data_loaders.PredictionsFromXarray(ds=p0012, variables=['2m_temperature'])
  l0618 = data_loaders.PredictionsFromXarray(ds=p0618, variables=['2m_temperature'])
  latency = np.timedelta64(6, 'h')
  multi = data_loaders.MultipleConstantLatencyWrapper([data_loaders.XarrayConstantLatencyWrapper(l0012, latency=latency),
                                                       data_loaders.XarrayConstantLatencyWrapper(l0618, latency=latency)])
  init_times, lead_times = _t('2020-01-02T00', '2020-01-02T06'), _h([6, 12])
  out = multi.load_chunk(init_times, lead_times)['2m_temperature']
  np.testing.assert_array_equal(out['init_time'].values, init_times)
  for i, (ai, al, src) in enumerate([(_t('2020-01-01T18'), _h([12, 18]), l0618), (_t('2020-01-02T00'), _h([12, 18]), l0012)]):
    np.testing.assert_array_equal(out.isel(init_time=[i]).values, src.load_chunk(ai, al)['2m_temperature'].values)
  with pytest.raises(NotImplementedError):
    multi._load_chunk_from_source(init_times, lead_times)


def test_multiple_latency_wrappers_tie_breaking():
  l1 = data_loaders.PredictionsFromXarray(ds=_forecasts('2020-01-01T00', '2020-01-02T00', 12, 24, 1, offset=1.0))
  l2 = data_loaders.PredictionsFromXarray(ds=_forecasts('2020-01-01T00', '2020-01-02T00', 12, 24, 1, offset=2.0))
  multi = data_loaders.MultipleConstantLatencyWrapper([
      data_loaders.XarrayConstantLatencyWrapper(l1, latency=np.timedelta64(6, 'h')),
      data_loaders.XarrayConstantLatencyWrapper(l2, latency=np.timedelta64(12, 'h'))])
  # query 13 UTC: 6 h latency -> issued 06 (nominal 00); 12 h latency -> issued 12 (nominal 00): same nominal init, the
  # larger latency wins
  out = multi.load_chunk(_t('2020-01-01T13'), _h([6]))['2m_temperature']
  want = l2.load_chunk(_t('2020-01-01T00'), _h([19]))['2m_temperature']
  np.testing.assert_array_equal(out.values, want.values)
  assert (out.values == 2.0).all()


def test_multiple_latency_wrappers_with_missing_init_time():
  loader = data_loaders.PredictionsFromXarray(ds=_forecasts('2020-01-01T00', '2020-01-02T00', 24, 12, 1, seed=23))
  multi = data_loaders.MultipleConstantLatencyWrapper([
      data_loaders.XarrayConstantLatencyWrapper(loader, latency=np.timedelta64(6, 'h')),
      data_loaders.XarrayConstantLatencyWrapper(loader, latency=np.timedelta64(1, 'h'))])
  out = multi.load_chunk(_t('2020-01-01T05'), _h([1]))['2m_temperature']       # only the 1 h latency has issued anything
  np.testing.assert_array_equal(out.values, loader.load_chunk(_t('2020-01-01T00'), _h([6]))['2m_temperature'].values)
  with pytest.raises(ValueError, match='No available init time found for init time'):
    multi.load_chunk(_t('2020-01-01T00'), _h([1]))


def test_latency_wrapper_keeps_the_wrapped_loaders_options():
  prediction = _forecasts('2020-01-01T00', '2020-01-03T00', 12, 30, 6, seed=24)
  prediction['2m_temperature'].values[3, 1, 2] = np.nan                          # lead 18 h of the 12 UTC run
  loader = data_loaders.PredictionsFromXarray(ds=prediction, add_nan_mask=True, process_chunk_fn=lambda c: c)
  wrapped = data_loaders.XarrayConstantLatencyWrapper(loader, latency=np.timedelta64(6, 'h'))
  out = wrapped.load_chunk(_t('2020-01-02T00'), _h([6, 12]))['2m_temperature']  # served from 01T12 at leads 18, 24
  mask = np.asarray(out.coords['mask'].values)
  np.testing.assert_array_equal(mask, ~np.isnan(out.values))
  assert (~mask).sum() == 1


def test_tensor_payloads_are_gathered_where_they_live():
  """A dataset whose payloads are torch tensors (what a field resident in HBM is) stays a tensor through every loader: the
  selections are index gathers on the tensor's device, no host round trip."""
  torch = pytest.importorskip('torch')
  target = mock_data.mock_target_data(time_start='2015-01-01T00', time_stop='2017-01-01T00', variables_3d=[], random=True,
                                      seed=30, dtype=np.float32)
  host = target['2m_temperature']
  dev = xr.Dataset({'2m_temperature': xr.DataArray(torch.from_numpy(np.ascontiguousarray(host.values)), dims=host.dims,
                                                   coords={d: host[d].values for d in host.dims})})
  init_times, lead_times = _inits('2016-12-28T00', '2016-12-30T00'), _h([0, 24, 48])
  for make in (lambda ds: data_loaders.TargetsFromXarray(ds=ds),
               lambda ds: data_loaders.PersistenceFromXarray(ds=ds),
               lambda ds: data_loaders.ProbabilisticClimatologyFromXarray(ds=ds, start_year=2015, end_year=2015),
               lambda ds: data_loaders.XarrayConstantLatencyWrapper(
                   data_loaders.PersistenceFromXarray(ds=ds), latency=np.timedelta64(24, 'h'), init_time_dim='valid_time')):
    a = make(target).load_chunk(init_times, lead_times)['2m_temperature']
    b = make(dev).load_chunk(init_times, lead_times)['2m_temperature']
    assert xr._is_torch(b.data) and not xr._is_torch(a.data), make  # pylint: disable=protected-access
    assert a.dims == b.dims
    np.testing.assert_array_equal(np.asarray(b.values), a.values)
    for d in a.dims:
      np.testing.assert_array_equal(b[d].values, a[d].values)


def test_mock_datasets_with_the_reference_signatures():
  """weatherbenchX/test_utils.py:27-104: the frames the reference's own tests build (e.g. latency_wrappers_test.py:24-33)."""
  from weatherbenchx_amd import test_utils  # pylint: disable=g-import-not-at-top
  ds = test_utils.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-04T00', time_resolution=np.timedelta64(12, 'h'),
                                       lead_start='0 hours', lead_stop='30 hours', lead_resolution='6 hours', random=True, seed=1)
  z = ds['geopotential']
  assert z.dims == ('prediction_timedelta', 'time', 'latitude', 'longitude', 'level') and z.shape == (6, 6, 19, 36, 3)
  np.testing.assert_array_equal(z['prediction_timedelta'].values, (np.arange(6) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'))
  assert z['time'].values[-1] == np.datetime64('2020-01-03T12', 'ns') and z.values.dtype == np.float64 and 0 <= z.values.min() < z.values.max() < 1
  t = test_utils.mock_target_data(variables_3d=[], ensemble_size=3, spatial_resolution_in_degrees=30.0, time_resolution='6 hours',
                                  time_start='2020-01-01', time_stop='2020-01-02')
  assert list(t) == ['2m_temperature'] and t['2m_temperature'].dims == ('time', 'latitude', 'longitude', 'realization')
  assert t['2m_temperature'].shape == (4, 7, 12, 3) and t['2m_temperature'].values.dtype == np.float32 and not t['2m_temperature'].values.any()
  # ... and they feed the loaders as the reference's fixtures do
  loader = data_loaders.PredictionsFromXarray(ds=ds, variables=['2m_temperature'])
  chunk = loader.load_chunk(np.array(['2020-01-02T00'], dtype='datetime64[ns]'), np.array([6, 12], dtype='timedelta64[h]'))
  assert chunk['2m_temperature'].sizes['lead_time'] == 2
