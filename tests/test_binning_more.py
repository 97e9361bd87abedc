"""The coordinate / time binnings (weatherbenchX/binning.py:204-705): mask construction against the reference's own known
answers (binning_test.py:62-400, restated on synthetic frames: the reference's fixtures need pandas / parquet loaders), and
through the Aggregator against the float64 oracle."""
import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic

H = np.timedelta64(1, 'h').astype('timedelta64[ns]')


def _gridded(times=None, leads=None, nlat=19, nlon=36, seed=0, values=False):
  rng = np.random.default_rng(seed)
  coords, dims = {}, []
  if leads is not None:
    coords['prediction_timedelta'] = leads
    dims.append('prediction_timedelta')
  if times is not None:
    coords['time'] = times
    dims.append('time')
  coords['latitude'] = np.linspace(-90, 90, nlat)
  coords['longitude'] = np.linspace(0, 360, nlon, endpoint=False)
  dims += ['latitude', 'longitude']
  shape = tuple(len(coords[d]) for d in dims)
  data = rng.normal(size=shape) if values else np.zeros(shape, np.float32)
  return xr.DataArray(data, dims=tuple(dims), coords=coords)


def _sparse(n=200, seed=1):
  """A station-like statistic: one `index` dim with non-dimension coordinates (lead_time, stationName), as the reference's sparse
  loaders hand on (binning_test.py:62-97, 185-265)."""
  rng = np.random.default_rng(seed)
  leads = (rng.integers(1, 7, n) * H)
  names = np.array([f'ST{int(i):03d}' for i in rng.integers(0, 40, n)])
  return xr.DataArray(rng.normal(size=n), dims=('index',),
                      coords={'index': np.arange(n), 'lead_time': (('index',), leads), 'stationName': (('index',), names)})


def test_by_exact_coord():
  stat = _sparse()
  mask = binning.ByExactCoord(coord='lead_time').create_bin_mask(stat)
  assert mask.dims == ('lead_time', 'index')
  np.testing.assert_array_equal(np.asarray(mask.coords['lead_time'].values), np.unique(np.asarray(stat.coords['lead_time'].values)))
  assert np.asarray(mask.values).sum(axis=0).tolist() == [1] * stat.shape[0]  # every point in exactly one bin
  mask = binning.ByExactCoord(coord='stationName', add_global_bin=True).create_bin_mask(stat)
  names = np.unique(np.asarray(stat.coords['stationName'].values))
  assert mask.shape[0] == len(names) + 1 and np.asarray(mask.coords['stationName'].values)[0] == 'global'
  assert np.asarray(mask.values)[0].all()
  empty = stat.isel(index=np.array([], dtype=int))
  assert np.asarray(binning.ByExactCoord(coord='stationName', add_global_bin=True).create_bin_mask(empty).values).size == 0
  with pytest.raises(AssertionError, match='reduce_dims'):
    binning.ByExactCoord(coord='index').create_bin_mask(stat)


def test_by_time_unit_datetime_and_timedelta():
  times = np.datetime64('2020-01-01T00', 'ns') + np.arange(12) * H
  mask = binning.ByTimeUnit('hour', 'time').create_bin_mask(_gridded(times=times))
  np.testing.assert_array_equal(np.asarray(mask.coords['time_hour'].values), np.arange(0, 12))
  assert mask.dims == ('time_hour', 'time') and np.array_equal(np.asarray(mask.values), np.eye(12, dtype=bool))
  for unit, res, stop in (('second', 1, 6), ('minute', 60, 360), ('hour', 3600, 6 * 3600), ('day', 86400, 6 * 86400),
                          ('week', 7 * 86400, 42 * 86400), ('year', 365 * 86400, 6 * 365 * 86400), ('hour', 900, 6 * 3600)):
    leads = (np.arange(0, stop + res, res) * np.timedelta64(1, 's')).astype('timedelta64[ns]')
    mask = binning.ByTimeUnit(unit, 'prediction_timedelta').create_bin_mask(_gridded(leads=leads, nlat=3, nlon=4))
    np.testing.assert_array_equal(np.asarray(mask.coords[f'prediction_timedelta_{unit}'].values), np.arange(0, 7))
  with pytest.raises(ValueError, match='Unsupported unit for timedelta'):
    binning.ByTimeUnit('month', 'prediction_timedelta').create_bin_mask(_gridded(leads=np.arange(3) * H, nlat=3, nlon=4))
  # calendar fields of a datetime coordinate
  t = np.array(['2019-12-31T23:59:58', '2020-02-29T06:30:15', '2021-03-01T00:00:00'], dtype='datetime64[ns]')
  f = binning._extract_time_unit  # pylint: disable=protected-access
  assert f(t, 'year').tolist() == [2019, 2020, 2021] and f(t, 'month').tolist() == [12, 2, 3] and f(t, 'day').tolist() == [31, 29, 1]
  assert f(t, 'dayofyear').tolist() == [365, 60, 60] and f(t, 'hour').tolist() == [23, 6, 0]
  assert f(t, 'minute').tolist() == [59, 30, 0] and f(t, 'second').tolist() == [58, 15, 0]
  assert f(t, 'dayofweek').tolist() == [1, 5, 0]  # Tuesday, Saturday, Monday


def test_by_time_unit_sets():
  times = np.datetime64('2020-01-01T00', 'ns') + np.arange(4) * 6 * H
  b = binning.ByTimeUnitSets(sets={'00/12': [0, 12], '06/18': [6, 18]}, unit='hour', dim='time', bin_dim_name='init_hour_sets')
  mask = b.create_bin_mask(_gridded(times=times))
  np.testing.assert_array_equal(np.asarray(mask.coords['init_hour_sets'].values), ['00/12', '06/18'])
  np.testing.assert_array_equal(np.asarray(mask.values), [[True, False, True, False], [False, True, False, True]])
  leads = np.arange(0, 30, 6) * H
  b = binning.ByTimeUnitSets(sets={'short': [0, 6], 'long': [12, 18, 24]}, unit='hour', dim='prediction_timedelta')
  assert b.bin_dim_name == 'prediction_timedelta_hour_sets'
  mask = b.create_bin_mask(_gridded(leads=leads, nlat=3, nlon=4))
  assert np.asarray(mask.values).sum(axis=1).tolist() == [2, 3]
  b = binning.ByTimeUnitSets(sets={'00/12': [0, 12]}, unit='hour', dim='time', add_global_bin=True)
  mask = b.create_bin_mask(_gridded(times=times))
  assert np.asarray(mask.coords[b.bin_dim_name].values).tolist() == ['00/12', 'global'] and np.asarray(mask.values)[1].all()


@pytest.mark.parametrize('unit, bins, expected', [('second', None, np.arange(60)), ('second', [0, 15, 30, 45], [0, 15, 30, 45]),
                                                  ('minute', None, np.arange(60)), ('minute', [0, 30], [0, 30]),
                                                  ('hour', None, np.arange(24)), ('hour', [0, 6, 12, 18], [0, 6, 12, 18])])
def test_by_time_unit_from_seconds(unit, bins, expected):
  secs = np.arange(0, 24 * 3600 + 1, 7)
  stat = xr.DataArray(np.zeros((secs.size, 2)), dims=('prediction_timedelta', 'latitude'),
                      coords={'prediction_timedelta': secs * np.timedelta64(1, 's').astype('timedelta64[ns]'), 'latitude': np.array([0., 10.]),
                              'prediction_timedelta_sec': (('prediction_timedelta',), secs.astype(np.float64))})
  mask = binning.ByTimeUnitFromSeconds(unit, 'prediction_timedelta_sec', bins=bins).create_bin_mask(stat)
  np.testing.assert_array_equal(np.asarray(mask.coords[f'prediction_timedelta_sec_{unit}'].values), expected)
  div = {'second': 1, 'minute': 60, 'hour': 3600}[unit]
  np.testing.assert_array_equal(np.asarray(mask.values), (secs // div)[None, :] == np.asarray(expected)[:, None])
  with pytest.raises(ValueError, match='Unsupported unit'):
    binning.ByTimeUnitFromSeconds('day', 'prediction_timedelta_sec').create_bin_mask(stat)


def test_by_coord_bins_and_by_sets():
  stat = _sparse()
  edges = np.arange(1, 8) * H
  mask = binning.ByCoordBins('lead_time', edges).create_bin_mask(stat)
  assert mask.dims == ('lead_time', 'index') and (np.asarray(mask.values).mean(axis=1) > 0).all()
  assert np.asarray(mask.values).sum(axis=0).tolist() == [1] * stat.shape[0]
  np.testing.assert_array_equal(np.asarray(mask.coords['lead_time'].values), edges[:-1])
  mask = binning.ByCoordBins('lead_time', edges, add_global_bin=True).create_bin_mask(stat)
  assert mask.shape[0] == 7 and np.asarray(mask.coords['lead_time'].values)[-1] == 'global' and np.asarray(mask.values)[-1].all()
  # with a global bin the labels are `str(edge)` (binning.py:605)
  assert np.asarray(mask.coords['lead_time'].values)[:-1].tolist() == [str(e) for e in edges[:-1]]
  fl = binning.ByCoordBins('lead_time', np.array([0.0, 2.5, 5.0]), add_global_bin=True).create_bin_mask(
      stat.assign_coords(lead_time=(('index',), np.linspace(0.0, 4.9, stat.shape[0]))))
  assert np.asarray(fl.coords['lead_time'].values).tolist() == ['0.0', '2.5', 'global']
  names = np.asarray(stat.coords['stationName'].values)
  uniq = np.unique(names)
  b = binning.BySets({'set1': uniq[:10], 'set2': uniq[10:20], 'scalar_set': uniq[0], 'empty_set': [], 'wrong_set': [1, 2, 3, 4]},
                     coord_name='stationName', bin_dim_name='station_subset', add_global_bin=True)
  mask = b.create_bin_mask(stat)
  labels = np.asarray(mask.coords['station_subset'].values).tolist()
  assert labels == ['set1', 'set2', 'scalar_set', 'empty_set', 'wrong_set', 'global']
  count = dict(zip(labels, np.asarray(mask.values).sum(axis=1).tolist()))
  assert count['set1'] == np.isin(names, uniq[:10]).sum() >= 10 and count['set2'] == np.isin(names, uniq[10:20]).sum() >= 10
  assert count['scalar_set'] == (names == uniq[0]).sum() and count['empty_set'] == 0 and count['wrong_set'] == 0 and count['global'] == len(names)
  b = binning.BySets({'a': uniq[:3]}, coord_name='stationName', bin_dim_name='subset', add_set_complements=True)
  mask = b.create_bin_mask(stat)
  assert np.asarray(mask.coords['subset'].values).tolist() == ['a', 'not_in_a'] and (np.asarray(mask.values).sum(axis=0) == 1).all()
  with pytest.raises(ValueError, match='different from coord_name'):
    binning.BySets({'a': [1]}, coord_name='x', bin_dim_name='x')


@pytest.mark.parametrize('degrees, lat_range, nbins', [(10, (-90, 90), 18), (30, (-90, 90), 6), (20, (0, 60), 3)])
def test_latitude_bins(degrees, lat_range, nbins):
  stat = _gridded(times=np.datetime64('2020-01-01', 'ns') + np.arange(1) * H)
  mask = binning.LatitudeBins(degrees, lat_range).create_bin_mask(stat)
  labels = np.asarray(mask.coords['latitude_bins'].values)
  assert labels.shape[0] == nbins and (labels >= lat_range[0]).all() and (labels < lat_range[1]).all()
  assert mask.shape == (nbins,) + stat.shape
  lat_val = 25 if lat_range[0] <= 25 < lat_range[1] else (lat_range[0] + lat_range[1]) / 2
  lat = np.asarray(stat.coords['latitude'].values)
  i = int(np.argmin(np.abs(lat - lat_val)))
  assert np.asarray(mask.values)[int((lat[i] - lat_range[0]) // degrees), 0, i, 0]


@pytest.mark.parametrize('degrees, lon_range, nbins, test_lon', [(10, (0, 360), 36, 10), (30, (0, 360), 12, 150), (60, (-180, 180), 6, 0),
                                                               (90, (270, 360), 1, 300)])
def test_longitude_bins(degrees, lon_range, nbins, test_lon):
  stat = _gridded(times=np.datetime64('2020-01-01', 'ns') + np.arange(1) * H)
  mask = binning.LongitudeBins(degrees, lon_range).create_bin_mask(stat)
  labels = np.asarray(mask.coords['longitude_bins'].values)
  assert labels.shape[0] == nbins and mask.shape == (nbins,) + stat.shape
  if lon_range == (-180, 180):
    assert 0 in labels
  lon = np.asarray(stat.coords['longitude'].values)
  j = int(np.argmin(np.abs(lon - test_lon)))
  k = int((lon[j] - lon_range[0]) // degrees) if lon[j] >= lon_range[0] else int((lon[j] + 360 - lon_range[0]) // degrees)
  assert np.asarray(mask.values)[k, 0, 0, j]


def test_time_binnings_through_the_aggregator(backend):
  """RMSE binned by the hour of the initialisation (ByTimeUnit), by named hour sets (ByTimeUnitSets) and by latitude bands on top of
  GridAreaWeighting: sums and weights of every bin against the float64 oracle; a sparse statistic binned by exact lead time."""
  rng = np.random.default_rng(3)
  times = np.datetime64('2020-01-01T00', 'ns') + np.arange(8) * 6 * H
  nlat, nlon = 13, 24
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  pv, tv = rng.normal(size=(8, nlat, nlon)), rng.normal(size=(8, nlat, nlon))
  dims = ('init_time', 'latitude', 'longitude')
  cs = {'init_time': times, 'latitude': lat, 'longitude': lon}
  p, t = {'v': xr.DataArray(pv, dims=dims, coords=cs)}, {'v': xr.DataArray(tv, dims=dims, coords=cs)}
  metrics = {'mse': deterministic.MSE()}
  hours = (times - times.astype('datetime64[D]')).astype('timedelta64[h]').astype(int)
  w = (O.grid_area_weights(lat), ('latitude',))
  se = (pv - tv) ** 2
  cases = [
      (binning.ByTimeUnit('hour', 'init_time'), 'init_time_hour', hours[None, :] == np.unique(hours)[:, None], ('init_time_hour', 'init_time')),
      (binning.ByTimeUnitSets({'00/12': [0, 12], 'six': 6}, 'hour', 'init_time'), 'init_time_hour_sets',
       np.stack([np.isin(hours, [0, 12]), hours == 6]), ('init_time_hour_sets', 'init_time')),
      (binning.LatitudeBins(30), 'latitude_bins',
       np.stack([(lat >= a) & (lat <= a + 30) for a in np.arange(-90, 90, 30)]), ('latitude_bins', 'latitude')),
  ]
  for b, bin_dim, want_mask, mdims in cases:
    agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], bin_by=[b])
    state = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
    sws, sw, out_dims = O.aggregate(se, dims, ['init_time', 'latitude', 'longitude'], weights=[w], bin_masks=[(bin_dim, want_mask, mdims)])
    got_s = np.asarray(state.sum_weighted_statistics['SquaredError']['v'].transpose(*out_dims).values)
    got_w = np.asarray(state.sum_weights['SquaredError']['v'].transpose(*out_dims).values)
    np.testing.assert_allclose(got_s, sws, rtol=1e-9, err_msg=bin_dim)
    np.testing.assert_allclose(got_w, sw, rtol=1e-12, err_msg=bin_dim)
  # sparse: one `index` dim, binned by the exact value of a non-dimension coordinate
  n = 300
  leads = rng.integers(1, 5, n) * H
  ps, ts = rng.normal(size=n), rng.normal(size=n)
  cs = {'index': np.arange(n), 'lead_time': (('index',), leads)}
  p, t = {'v': xr.DataArray(ps, dims=('index',), coords=cs)}, {'v': xr.DataArray(ts, dims=('index',), coords=cs)}
  agg = aggregation.Aggregator(reduce_dims=['index'], bin_by=[binning.ByExactCoord('lead_time')])
  vals = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t)).metric_values(metrics)['mse.v']
  want = [((ps - ts)[leads == u] ** 2).mean() for u in np.unique(leads)]
  np.testing.assert_allclose(np.asarray(vals.values), want, rtol=1e-9)
  np.testing.assert_array_equal(np.asarray(vals.coords['lead_time'].values), np.unique(leads))


def test_latitude_and_longitude_bands_of_station_data():
  """Sparse statistics carry latitude / longitude as coordinates over `index`; the bands broadcast along that dim, as
  `mask.broadcast_like(statistic)` does (binning.py:232-238, 277-284)."""
  rng = np.random.default_rng(11)
  n = 80
  lat, lon = rng.uniform(-90, 90, n), rng.uniform(-180, 180, n)
  stat = xr.DataArray(rng.normal(size=(3, n)), dims=('lead_time', 'index'),
                      coords={'lead_time': np.arange(3) * H, 'latitude': (('index',), lat), 'longitude': (('index',), lon)})
  mask = binning.LatitudeBins(30).create_bin_mask(stat)
  assert mask.dims == ('latitude_bins', 'lead_time', 'index') and mask.shape == (6, 3, n)
  for i, a in enumerate(range(-90, 90, 30)):
    np.testing.assert_array_equal(np.asarray(mask.values)[i], np.broadcast_to((lat >= a) & (lat <= a + 30), (3, n)))
  mask = binning.LongitudeBins(90).create_bin_mask(stat)
  assert mask.shape == (4, 3, n)
  m360 = np.mod(lon, 360)
  for i, a in enumerate(range(0, 360, 90)):
    want = (m360 >= a) & (m360 <= a + 90) if a + 90 < 360 else (m360 >= a) | (m360 <= 0)
    np.testing.assert_array_equal(np.asarray(mask.values)[i, 0], want)


def test_time_units_of_the_dt_accessor_beyond_the_calendar_fields():
  """the reference forwards any integer field of `.dt` (binning.py: `getattr(coord.dt, unit)`): quarter, ISO week,
  days_in_month, is_leap_year -- checked against the standard library."""
  import calendar
  import datetime
  from weatherbenchx_amd import binning
  t = np.array(['2020-01-01T05', '2021-01-03T00', '2020-12-31T23', '2024-02-29T12', '2019-12-30T00', '2022-07-15T06'], dtype='datetime64[ns]')
  py = [datetime.datetime.fromisoformat(str(x)[:19]) for x in t]
  assert list(binning._extract_time_unit(t, 'week')) == [d.isocalendar()[1] for d in py]
  assert list(binning._extract_time_unit(t, 'weekofyear')) == [d.isocalendar()[1] for d in py]
  assert list(binning._extract_time_unit(t, 'quarter')) == [(d.month - 1) // 3 + 1 for d in py]
  assert list(binning._extract_time_unit(t, 'days_in_month')) == [calendar.monthrange(d.year, d.month)[1] for d in py]
  assert list(binning._extract_time_unit(t, 'is_leap_year')) == [int(calendar.isleap(d.year)) for d in py]
  with pytest.raises(ValueError, match='season'):
    binning._extract_time_unit(t, 'season')
