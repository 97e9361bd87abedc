"""The small input transforms (weatherbenchX/metrics/wrappers.py:50-89, 214-267, 587-645, 745-808, 1072-1134): the reference's own
known answers (wrappers_test.py:27-130, 227-245) on this package's labeled arrays, and wrapped metrics through the Aggregator against
the float64 oracle."""
import numpy as np
import pytest

import mock_data
from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import wrappers


def _target():
  return mock_data.mock_target_data(random=True, seed=0, time_start='2020-01-01', time_stop='2020-01-09')


def test_continuous_to_binary_constant_and_iterable_thresholds():
  x = _target()['geopotential']
  ctb = wrappers.ContinuousToBinary(which='both', threshold_value=0.5, threshold_dim='threshold', unique_name_suffix='test')
  y = ctb.transform_fn(x)
  np.testing.assert_array_equal(np.asarray(y.coords['threshold'].values), [0.5])
  np.testing.assert_array_equal(np.asarray(y.sel(threshold=0.5, drop=True).values), np.asarray(x.values) > 0.5)
  assert ctb.unique_name_suffix == 'threshold=test'
  ts = [0.2, 0.7]
  ctb = wrappers.ContinuousToBinary(which='both', threshold_value=ts, threshold_dim='threshold')
  y = ctb.transform_fn(x)
  np.testing.assert_array_equal(np.asarray(y.coords['threshold'].values), ts)
  for t in ts:
    np.testing.assert_array_equal(np.asarray(y.sel(threshold=t, drop=True).values), np.asarray(x.values) > t)
  assert ctb.unique_name_suffix == 'threshold=0.2,0.7' and y.dtype == np.float32
  # NaN stays NaN
  xn = x.copy()
  vals = np.array(xn.values)
  vals[0, 0, 0, 0] = np.nan
  yn = ctb.transform_fn(xr.DataArray(vals, dims=x.dims, coords={d: x.coords[d] for d in x.dims}, name='geopotential'))
  assert np.isnan(np.asarray(yn.values)[0, 0, 0, 0]).all() and np.isfinite(np.asarray(yn.values)).sum() == yn.values.size - 2


def test_continuous_to_binary_labeled_thresholds():
  ds = _target()
  q = [0.25, 0.75]
  for labeled in ('dataarray', 'dataset'):
    per_var = {}
    for var in ('geopotential', '2m_temperature'):
      v = ds[var]
      tq = np.quantile(np.asarray(v.values), q, axis=v.dims.index('time'))
      rest = tuple(d for d in v.dims if d != 'time')
      per_var[var] = xr.DataArray(tq, dims=('threshold',) + rest, coords={'threshold': np.array(q), **{d: v.coords[d] for d in rest}}, name=var)
    thr = per_var['geopotential'] if labeled == 'dataarray' else xr.Dataset(per_var)
    with pytest.raises(ValueError, match='unique_name_suffix must be provided'):
      wrappers.ContinuousToBinary(which='both', threshold_value=thr, threshold_dim='threshold')
    ctb = wrappers.ContinuousToBinary(which='both', threshold_value=thr, threshold_dim='threshold', unique_name_suffix='test')
    for var in (('geopotential',) if labeled == 'dataarray' else ('geopotential', '2m_temperature')):
      x = ds[var]
      y = ctb.transform_fn(x)
      np.testing.assert_array_equal(np.asarray(y.coords['threshold'].values), q)
      for k, t in enumerate(q):
        want = np.asarray(x.values) > np.expand_dims(np.asarray(per_var[var].values)[k], x.dims.index('time'))
        np.testing.assert_array_equal(np.asarray(y.sel(threshold=t).transpose(*x.dims).values), want)


def test_inline_relu_rename_select():
  x = _target()['geopotential'] - 0.5
  y = wrappers.Inline('both', lambda da: -da, 'negate').transform_fn(x)
  np.testing.assert_array_equal(np.asarray(y.values), -np.asarray(x.values))
  vals = np.array(x.values)
  vals[1, 2, 3, 0] = np.nan
  xn = xr.DataArray(vals, dims=x.dims, coords={d: x.coords[d] for d in x.dims})
  relu = wrappers.ReLU('both')
  got = np.asarray(relu.transform_fn(xn).values)
  want = np.where(np.isnan(vals), np.nan, np.maximum(vals, 0))
  np.testing.assert_array_equal(got, want)
  assert relu.unique_name_suffix == 'relu'
  r = wrappers.Rename('both', {'level': 'pressure'})
  assert r.transform_fn(x).dims == tuple('pressure' if d == 'level' else d for d in x.dims) and r.unique_name_suffix == "rename_{'level': 'pressure'}"
  sel = wrappers.Select('both', sel={'level': 500}, isel={'time': slice(0, 3)})
  got = sel.transform_fn(x)
  assert 'level' not in got.dims and got.sizes['time'] == 3
  np.testing.assert_array_equal(np.asarray(got.values), np.asarray(x.values)[:3, :, :, list(np.asarray(x.coords['level'].values)).index(500)])
  assert sel.unique_name_suffix == ("select_self._isel={'time': slice(0, 3, None)}_self._isel_kwargs={}_self._sel={'level': 500}_self._sel_kwargs={}")
  with pytest.raises(ValueError, match='Invalid value for `which`'):
    wrappers.ReLU('nobody')


def test_wrapped_metrics_through_the_aggregator(backend):
  """MSE of exceedance indicators (the Brier-type score of two deterministic fields) under GridAreaWeighting, RMSE of one level
  picked with Select, and SubselectVariables: against the float64 oracle."""
  rng = np.random.default_rng(2)
  nt, nlat, nlon, nlev = 4, 13, 24, 3
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  dims = ('time', 'level', 'latitude', 'longitude')
  cs = {'time': np.datetime64('2020-01-01', 'ns') + np.arange(nt) * np.timedelta64(24, 'h'), 'level': np.array([500, 700, 850]), 'latitude': lat, 'longitude': lon}
  pv, tv = rng.random((nt, nlev, nlat, nlon)), rng.random((nt, nlev, nlat, nlon))
  pv[0, 1, 2, 3] = np.nan
  p = {'z': xr.DataArray(pv, dims=dims, coords=cs, name='z'), 'q': xr.DataArray(pv * 2, dims=dims, coords=cs, name='q')}
  t = {'z': xr.DataArray(tv, dims=dims, coords=cs, name='z'), 'q': xr.DataArray(tv * 2, dims=dims, coords=cs, name='q')}
  ts = [0.3, 0.6]
  metrics = {
      'brier': wrappers.WrappedMetric(deterministic.MSE(), [wrappers.ContinuousToBinary('both', ts, 'threshold')]),
      'rmse500': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.Select('both', sel={'level': 500})]),
      'mse_z': wrappers.SubselectVariables(deterministic.MSE(), ['z']),
  }
  agg = aggregation.Aggregator(reduce_dims=['time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], skipna=True)
  out = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t)).metric_values(metrics)
  assert set(out) == {'brier.z', 'brier.q', 'rmse500.z', 'rmse500.q', 'mse_z.z'}
  w = (O.grid_area_weights(lat), ('latitude',))

  def mean(stat, sdims):
    sws, sw, od = O.aggregate(stat, sdims, ['time', 'latitude', 'longitude'], weights=[w], skipna=True)
    return sws / sw, od
  for k, th in enumerate(ts):
    bp = np.where(np.isnan(pv), np.nan, (pv > th).astype(float))
    want, od = mean((bp - (tv > th)) ** 2, dims)
    np.testing.assert_allclose(np.asarray(out['brier.z'].sel(threshold=th).transpose(*od).values), want, rtol=1e-6)
  want, od = mean((pv[:, 0] - tv[:, 0]) ** 2, ('time', 'latitude', 'longitude'))
  np.testing.assert_allclose(np.asarray(out['rmse500.z'].values), np.sqrt(want), rtol=1e-6)
  want, od = mean((pv - tv) ** 2, dims)
  np.testing.assert_allclose(np.asarray(out['mse_z.z'].transpose(*od).values), want, rtol=1e-6)


@pytest.mark.parametrize('skipna', [True, False])
@pytest.mark.parametrize('quantiles', [0.5, [0.1, 0.5, 0.9]])
def test_ensemble_quantiles(skipna, quantiles):
  """wrappers_test.py:151-190: quantiles over `realization` with NaNs in one latitude row, against numpy's (nan)quantile."""
  ds = mock_data.mock_target_data(random=True, seed=1, ensemble_size=3, time_start='2020-01-01', time_stop='2020-01-04')
  x = ds['geopotential'].isel(latitude=slice(0, 2), longitude=slice(0, 2), level=0)
  vals = np.array(x.values)
  vals[:, 0] = np.nan
  x = xr.DataArray(vals, dims=x.dims, coords={d: x.coords[d] for d in x.dims}, name='geopotential')
  eq = wrappers.EnsembleQuantiles('both', quantiles, quantile_dim='my_quantile', ensemble_dim='realization', skipna=skipna)
  y = eq.transform_fn(x)
  qs = quantiles if isinstance(quantiles, list) else [quantiles]
  assert y.dims == ('my_quantile',) + tuple(d for d in x.dims if d != 'realization')
  np.testing.assert_array_equal(np.asarray(y.coords['my_quantile'].values), qs)
  import warnings
  with warnings.catch_warnings():
    warnings.simplefilter('ignore', RuntimeWarning)
    want = (np.nanquantile if skipna else np.quantile)(vals, qs, axis=x.dims.index('realization'))
  np.testing.assert_array_equal(np.asarray(y.values), want)
  with pytest.raises(ValueError, match='already has a `quantile` dimension'):
    eq.transform_fn(xr.DataArray(np.zeros((2, 3)), dims=('quantile', 'realization')))
  plain = xr.DataArray(np.zeros(3), dims=('time',))
  assert wrappers.EnsembleQuantiles('both', 0.5, ensemble_dim='realization', skip_if_ensemble_dim_missing=True).transform_fn(plain) is plain


def test_weibull_and_shift_along_new_dim():
  ds = mock_data.mock_target_data(random=True, seed=2, ensemble_size=3, time_start='2020-01-01', time_stop='2020-01-04')
  x = ds['geopotential']
  for skipna in (True, False):
    binary = wrappers.ContinuousToBinary('both', 0.5, 'threshold').transform_fn(x)
    y = wrappers.WeibullEnsembleToProbabilistic('predictions', ensemble_dim='realization', skipna=skipna).transform_fn(binary)
    want = (np.asarray(x.values) > 0.5).sum(axis=x.dims.index('realization')) / 4
    np.testing.assert_allclose(np.asarray(y.sel(threshold=0.5, drop=True).transpose(*[d for d in x.dims if d != 'realization']).values), want)
  with pytest.raises(AssertionError, match='Only predictions'):
    wrappers.WeibullEnsembleToProbabilistic('both')
  t = mock_data.mock_target_data(random=True, seed=3, time_start='2020-01-01', time_stop='2020-01-06')
  x = t['geopotential']
  y = wrappers.ShiftAlongNewDim('both', 0.5, 'threshold', 'shift_0.5').transform_fn(x)
  assert y.sizes['threshold'] == 1 and np.asarray(y.coords['threshold'].values).tolist() == [0.5]
  np.testing.assert_array_equal(np.asarray(y.sel(threshold=0.5, drop=True).transpose(*x.dims).values), np.asarray(x.values) + 0.5)
  y = wrappers.ShiftAlongNewDim('both', [0.2, 0.7], 'threshold', 'shift_two').transform_fn(x)
  for th in (0.2, 0.7):
    np.testing.assert_array_equal(np.asarray(y.sel(threshold=th, drop=True).transpose(*x.dims).values), np.asarray(x.values) + th)
  # per-variable fields from a Dataset that already carries the shift dim (here: time quantiles)
  per_var = {}
  for var in ('geopotential', '2m_temperature'):
    v = t[var]
    tq = np.quantile(np.asarray(v.values), [0.25, 0.75], axis=v.dims.index('time'))
    rest = tuple(d for d in v.dims if d != 'time')
    per_var[var] = xr.DataArray(tq, dims=('quantile',) + rest, coords={'quantile': np.array([0.25, 0.75]), **{d: v.coords[d] for d in rest}}, name=var)
  shift = wrappers.ShiftAlongNewDim('both', xr.Dataset(per_var), 'quantile', 'shift_q')
  y = shift.transform_fn(x)
  for k, qv in enumerate((0.25, 0.75)):
    want = np.asarray(x.values) + np.expand_dims(np.asarray(per_var['geopotential'].values)[k], x.dims.index('time'))
    np.testing.assert_allclose(np.asarray(y.sel(quantile=qv, drop=True).transpose(*x.dims).values), want)
  with pytest.raises(RuntimeError, match='Expected to find'):
    wrappers.ShiftAlongNewDim('both', xr.Dataset(per_var), 'threshold', 'x').transform_fn(x)


def test_ensemble_quantiles_of_a_tensor_payload_stay_a_tensor():
  """A payload that is a torch tensor (in HBM on a GPU box) goes through torch's quantile kernels and stays a tensor."""
  torch = pytest.importorskip('torch')
  rng = np.random.default_rng(4)
  v = rng.random((4, 3, 5)).astype(np.float32)
  v[0, :, 0] = np.nan
  x = xr.DataArray(torch.from_numpy(v.copy()), dims=('time', 'realization', 'latitude'))
  import warnings
  for skipna in (True, False):
    y = wrappers.EnsembleQuantiles('both', [0.1, 0.5], ensemble_dim='realization', skipna=skipna).transform_fn(x)
    assert xr._is_torch(y.data) and y.dims == ('quantile', 'time', 'latitude')  # pylint: disable=protected-access
    with warnings.catch_warnings():
      warnings.simplefilter('ignore', RuntimeWarning)
      want = (np.nanquantile if skipna else np.quantile)(v, [0.1, 0.5], axis=1)
    np.testing.assert_allclose(np.asarray(y.values), want, rtol=1e-6, equal_nan=True)


def _labeled_like(x, edges, dim):
  e = np.asarray(edges, float)
  return xr.DataArray(e, dims=[dim], coords={dim: e}) * xr.ones_like(x)


def test_continuous_to_bins_and_cdf():
  """wrappers_test.py:310-376: three right-inclusive bins from [-inf, 0.2, 0.7, inf], as values, as a field and as a Dataset of
  fields; non-monotonic edges refused; NaN propagated."""
  ds = mock_data.mock_target_data(random=True, seed=5, time_start='2020-01-01', time_stop='2020-01-05')
  x = ds['geopotential']
  edges = [-np.inf, 0.2, 0.7, np.inf]
  fields = {v: _labeled_like(ds[v], edges, 'bin_values') for v in ('geopotential', '2m_temperature')}
  for bin_values in (edges, fields['geopotential'], xr.Dataset({k: v.rename(k) for k, v in fields.items()})):
    ctb = wrappers.ContinuousToBins('both', bin_values, 'bin_values', unique_name_suffix='test')
    y = ctb.transform_fn(x)
    np.testing.assert_array_equal(np.asarray(y.coords['bin_values_left'].values), [-np.inf, 0.2, 0.7])
    np.testing.assert_array_equal(np.asarray(y.coords['bin_values_right'].values), [0.2, 0.7, np.inf])
    v = np.asarray(x.values)
    for k, want in enumerate((v <= 0.2, (v > 0.2) & (v <= 0.7), v > 0.7)):
      np.testing.assert_array_equal(np.asarray(y.isel(bin_values=k, drop=True).transpose(*x.dims).values), want)
  assert np.asarray(wrappers.ContinuousToBins('both', edges, 'b').transform_fn(x).coords['b'].values).tolist() == \
      ['-inf < p <= 0.20', '0.20 < p <= 0.70', '0.70 < p <= inf']
  with pytest.raises(ValueError, match='monotonically increasing'):
    wrappers.ContinuousToBins('both', [0.7, 0.2], 'bin_values').transform_fn(x)
  with pytest.raises(ValueError, match='unique_name_suffix must be provided'):
    wrappers.ContinuousToBins('both', fields['geopotential'], 'bin_values')
  nan_x = xr.DataArray(np.full(x.shape, np.nan), dims=x.dims, coords={d: x.coords[d] for d in x.dims}, name='geopotential')
  assert np.isnan(np.asarray(wrappers.ContinuousToBins('both', [0.2, 0.7], 'bin_values').transform_fn(nan_x).values)).all()
  for right in (True, False):
    cdf = wrappers.ContinuousToCDF('both', [0.25, 0.5], 'threshold', right_inclusive=right)
    y = cdf.transform_fn(x)
    v = np.asarray(x.values)
    for k, t in enumerate((0.25, 0.5)):
      np.testing.assert_array_equal(np.asarray(y.isel(threshold=k, drop=True).transpose(*x.dims).values), (v <= t) if right else (v < t))
    assert cdf.unique_name_suffix == f'ContinuousToCDF_threshold_0.25,0.5_right_inclusive_{right}'
  with pytest.raises(ValueError, match='must be an Iterable'):
    wrappers.compute_cdf(0.5, x, 'threshold', True)


def test_select_bin_thresholds_by_time_from_chunk():
  """wrappers_test.py:398-470: day-of-year thresholds picked by (init_time + lead_time), by valid_time, by a station chunk's `time`
  coordinate; plus valid_time- and (init_time, lead_time)-indexed thresholds and the pass-through cases."""
  doys = np.arange(1, 366)
  thr = xr.DataArray(np.arange(1, 366), dims=['dayofyear'], coords={'dayofyear': doys})
  inits = np.array(['2023-01-01', '2023-01-02'], dtype='datetime64[ns]')
  leads = np.array([24, 48], dtype='timedelta64[h]').astype('timedelta64[ns]')
  chunk = xr.DataArray(np.random.rand(2, 2), dims=['init_time', 'lead_time'], coords={'init_time': inits, 'lead_time': leads})
  got = wrappers.select_bin_thresholds_by_time_from_chunk(thr, chunk)
  assert got.dims == ('init_time', 'lead_time')
  np.testing.assert_array_equal(np.asarray(got.values), [[2, 3], [3, 4]])
  np.testing.assert_array_equal(np.asarray(got.coords['dayofyear'].values), [[2, 3], [3, 4]])
  np.testing.assert_array_equal(np.asarray(got.coords['init_time'].values), inits)
  valid = np.array(['2023-01-02', '2023-01-03'], dtype='datetime64[ns]')
  got = wrappers.select_bin_thresholds_by_time_from_chunk(thr, xr.DataArray(np.random.rand(2), dims=['valid_time'], coords={'valid_time': valid}))
  assert got.dims == ('valid_time',) and np.asarray(got.values).tolist() == [2, 3]
  station = xr.DataArray(np.random.rand(2), dims=['index'], coords={'time': (('index',), valid)})
  got = wrappers.select_bin_thresholds_by_time_from_chunk(thr, station)
  assert got.dims == ('index',) and np.asarray(got.values).tolist() == [2, 3]
  np.testing.assert_array_equal(np.asarray(got.coords['time'].values), valid)
  # thresholds by valid_time with a spatial dim, picked for an (init, lead) chunk
  vt = np.datetime64('2023-01-01', 'ns') + np.arange(6) * np.timedelta64(24, 'h')
  by_valid = xr.DataArray(np.arange(12.).reshape(6, 2), dims=['valid_time', 'latitude'], coords={'valid_time': vt, 'latitude': np.array([0., 10.])})
  got = wrappers.select_bin_thresholds_by_time_from_chunk(by_valid, chunk)
  assert got.dims == ('init_time', 'lead_time', 'latitude')
  np.testing.assert_array_equal(np.asarray(got.values)[..., 0], [[2, 4], [4, 6]])
  by_il = xr.DataArray(np.arange(12.).reshape(3, 4), dims=['init_time', 'lead_time'],
                       coords={'init_time': np.datetime64('2023-01-01', 'ns') + np.arange(3) * np.timedelta64(24, 'h'),
                               'lead_time': (np.arange(4) * 24).astype('timedelta64[h]').astype('timedelta64[ns]')})
  got = wrappers.select_bin_thresholds_by_time_from_chunk(by_il, chunk)
  np.testing.assert_array_equal(np.asarray(got.values), [[1, 2], [5, 6]])
  plain = xr.DataArray(np.arange(3.), dims=['threshold'], coords={'threshold': np.arange(3.)})
  assert wrappers.select_bin_thresholds_by_time_from_chunk(plain, chunk) is plain
  assert wrappers.select_bin_thresholds_by_time_from_chunk(thr, xr.DataArray(np.zeros(3), dims=['latitude'])) is thr
  with pytest.raises(KeyError, match='not all values found'):
    wrappers.select_bin_thresholds_by_time_from_chunk(by_valid.isel(valid_time=slice(0, 2)), chunk)
  # ... and a CDF against day-of-year thresholds that follow the chunk
  field = xr.DataArray(np.array([[1.5, 3.5], [2.5, 4.5]]), dims=['init_time', 'lead_time'], coords={'init_time': inits, 'lead_time': leads}, name='v')
  thr2 = xr.DataArray(np.stack([np.arange(1, 366) - 0.25, np.arange(1, 366) + 0.25]).astype(float), dims=['threshold', 'dayofyear'],
                      coords={'threshold': np.array([0, 1]), 'dayofyear': doys})
  cdf = wrappers.ContinuousToCDF('both', thr2, 'threshold', unique_name_suffix='doy').transform_fn(field)
  # thresholds at (init, lead): doy -+ 0.25 = [[1.75|2.25, 2.75|3.25], [2.75|3.25, 3.75|4.25]]
  np.testing.assert_array_equal(np.asarray(cdf.sel(threshold=0, drop=True).transpose('init_time', 'lead_time').values), [[1, 0], [1, 0]])
  np.testing.assert_array_equal(np.asarray(cdf.sel(threshold=1, drop=True).transpose('init_time', 'lead_time').values), [[1, 0], [1, 0]])


def test_stack_to_new_dimension_and_tiles():
  x = mock_data.mock_target_data(random=True, seed=6, time_start='2020-01-01', time_stop='2020-01-03')['geopotential']
  st = wrappers.StackToNewDimension('both', ['latitude', 'longitude'], 'latitude')
  y = st.transform_fn(x)
  n = x.sizes['latitude'] * x.sizes['longitude']
  assert y.dims == ('time', 'level', 'latitude') and y.sizes['latitude'] == n
  np.testing.assert_array_equal(np.asarray(y.coords['latitude'].values), np.arange(n))
  np.testing.assert_array_equal(np.asarray(y.values), np.asarray(x.transpose('time', 'level', 'latitude', 'longitude').values).reshape(x.sizes['time'], x.sizes['level'], n))
  assert st.unique_name_suffix == "stack_['latitude', 'longitude']_to_latitude"
  v = np.asarray(x.values)
  nlat, nlon = x.sizes['latitude'], x.sizes['longitude']
  ilat, ilon = x.dims.index('latitude'), x.dims.index('longitude')
  for wrap in (False, True):
    t = wrappers.Tile('both', window_size=3, wrap_longitude=wrap).transform_fn(x)
    assert t.dims == ('window',) + x.dims and t.sizes['window'] == 9
    assert t.sizes['latitude'] == nlat - 2 and t.sizes['longitude'] == (nlon if wrap else nlon - 2)
    np.testing.assert_array_equal(np.asarray(t.coords['latitude'].values), np.asarray(x.coords['latitude'].values)[1:-1])
    got = np.asarray(t.values)
    for w, (di, dj) in enumerate((i - 1, j - 1) for i in range(3) for j in range(3)):
      rolled = np.roll(np.roll(v, di, axis=ilat), dj, axis=ilon)
      sl = [slice(None)] * v.ndim
      sl[ilat] = slice(1, nlat - 1)
      if not wrap:
        sl[ilon] = slice(1, nlon - 1)
      np.testing.assert_array_equal(got[w], rolled[tuple(sl)])
  t5 = wrappers.construct_tiles(x, window_size=4)
  assert t5.sizes['window'] == 16 and t5.sizes['latitude'] == nlat - 3 and t5.sizes['longitude'] == nlon - 3
  assert wrappers.Tile('both', 3, 'w', True).unique_name_suffix == 'tiled_window_size_3_wrap_True_dim_w'


def test_tensor_payloads_take_the_same_paths():
  """The transforms on torch payloads (what a chunk in HBM is): same numbers, and the result stays a tensor."""
  torch = pytest.importorskip('torch')
  rng = np.random.default_rng(7)
  v = rng.random((2, 5, 6)).astype(np.float32)
  dims = ('time', 'latitude', 'longitude')
  cs = {'time': np.datetime64('2023-01-02', 'ns') + np.arange(2) * np.timedelta64(24, 'h'), 'latitude': np.linspace(-40, 40, 5), 'longitude': np.arange(6) * 60.0}
  xn = xr.DataArray(v.copy(), dims=dims, coords=cs, name='v')
  xt = xr.DataArray(torch.from_numpy(v.copy()), dims=dims, coords=cs, name='v')
  thr = xr.DataArray(torch.from_numpy(np.stack([np.arange(1, 366) / 400.0, np.arange(1, 366) / 300.0]).astype(np.float32)), dims=['threshold', 'dayofyear'],
                     coords={'threshold': np.array([0, 1]), 'dayofyear': np.arange(1, 366)})
  thr_n = xr.DataArray(np.asarray(thr.values), dims=thr.dims, coords={'threshold': np.array([0, 1]), 'dayofyear': np.arange(1, 366)})
  for make in (lambda: wrappers.Tile('both', 3, wrap_longitude=True), lambda: wrappers.StackToNewDimension('both', ['latitude', 'longitude'], 'point'),
               lambda: wrappers.ContinuousToBins('both', [-np.inf, 0.3, np.inf], 'bin'), lambda: wrappers.ReLU('both'),
               lambda: wrappers.ContinuousToBinary('both', [0.2, 0.6], 'threshold')):
    a, b = make().transform_fn(xn), make().transform_fn(xt)
    assert xr._is_torch(b.data) and a.dims == b.dims  # pylint: disable=protected-access
    np.testing.assert_allclose(np.asarray(b.values), np.asarray(a.values), rtol=1e-6, equal_nan=True)
  a = wrappers.ContinuousToCDF('both', thr_n, 'threshold', unique_name_suffix='d', enforce_monotonicity=False).transform_fn(xn)
  b = wrappers.ContinuousToCDF('both', thr, 'threshold', unique_name_suffix='d', enforce_monotonicity=False).transform_fn(xt)
  assert xr._is_torch(b.data)  # pylint: disable=protected-access
  np.testing.assert_array_equal(np.asarray(b.transpose(*a.dims).values), np.asarray(a.values))
