"""Chunk pre-processing (weatherbenchx_amd/interpolations.py): the cases of weatherbenchX/interpolations_test.py:24-308 restated,
and the interpolation itself against scipy.interpolate -- the library the reference reaches through `xarray.DataArray.interp`
(interpolations.py:89-113) -- orthogonally and pointwise, linear and nearest, with and without extrapolation, NaNs, descending
axes, host and tensor payloads; a loader that regrids its chunks to the targets feeding an evaluation."""
import numpy as np
import pytest
from scipy import interpolate as sci

from tests import mock_data
from weatherbenchx_amd import data_loaders
from weatherbenchx_amd import interpolations
from weatherbenchx_amd import xarray_lite as xr


def _grid(resolution, seed=None, **kw):
  return mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-02T00', time_resolution_hours=12,
                                        lead_stop_days=1, spatial_resolution_in_degrees=resolution, random=seed is not None,
                                        seed=seed, **kw)


# ---- interp against scipy ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('method', ['linear', 'nearest'])
@pytest.mark.parametrize('extrapolate', [True, False])
def test_orthogonal_interpolation_against_scipy(method, extrapolate):
  rng = np.random.default_rng(1)
  lat, lon = np.array([-60.0, -20.0, 5.0, 40.0, 75.0]), np.array([0.0, 30.0, 90.0, 180.0, 270.0, 330.0])
  values = rng.normal(size=(3, 5, 6))
  values[1, 2, 3] = np.nan
  da = xr.DataArray(values, dims=('time', 'latitude', 'longitude'), coords={'time': np.arange(3), 'latitude': lat, 'longitude': lon})
  new_lat, new_lon = np.array([-70.0, -60.0, -33.3, 5.0, 22.5, 80.0]), np.array([-10.0, 0.0, 15.0, 60.1, 300.0, 350.0])
  got = interpolations.interp(da, {'latitude': new_lat, 'longitude': new_lon}, method, extrapolate)
  assert got.dims == ('time', 'latitude', 'longitude')
  np.testing.assert_array_equal(got['latitude'].values, new_lat)
  pts = np.stack(np.meshgrid(new_lat, new_lon, indexing='ij'), axis=-1)
  for k in range(3):
    want = sci.interpn((lat, lon), values[k], pts, method=method, bounds_error=False, fill_value=None if extrapolate else np.nan)
    np.testing.assert_allclose(got.values[k], want, rtol=1e-12, atol=1e-12, equal_nan=True)
  # one dim only, descending source axis, a DataArray target over the dim itself
  down = da.isel(latitude=np.arange(5)[::-1])
  one = interpolations.interp(down, {'latitude': xr.DataArray(new_lat, dims=('latitude',))}, method, extrapolate)
  f = sci.interp1d(lat, values, axis=1, kind=method, bounds_error=False, fill_value='extrapolate' if extrapolate else np.nan)
  np.testing.assert_allclose(one.values, f(new_lat), rtol=1e-12, atol=1e-12, equal_nan=True)
  # a scalar target drops the dim and keeps the coordinate
  at = interpolations.interp(da, {'longitude': np.array(45.0)}, method, extrapolate)
  assert at.dims == ('time', 'latitude') and float(at.coords['longitude'].values) == 45.0


@pytest.mark.parametrize('method', ['linear', 'nearest'])
def test_pointwise_interpolation_against_scipy(method):
  rng = np.random.default_rng(2)
  lat, lon = np.linspace(-80, 80, 9), np.arange(0, 360, 30.0)
  values = rng.normal(size=(9, 2, 12)).astype(np.float32)               # the interpolated dims are not adjacent
  da = xr.DataArray(values, dims=('latitude', 'level', 'longitude'), coords={'latitude': lat, 'level': [500, 850], 'longitude': lon})
  n = 25
  slat, slon = rng.uniform(-95, 95, n), rng.uniform(-20, 350, n)
  stations = {'index': np.arange(n), 'elevation': (('index',), rng.uniform(0, 2000, n)), 'latitude': (('index',), slat),
              'longitude': (('index',), slon)}
  ref = xr.DataArray(np.zeros(n), dims=('index',), coords=stations)
  for extrapolate in (True, False):
    got = interpolations.interp(da, {'latitude': ref['latitude'], 'longitude': ref['longitude']}, method, extrapolate)
    assert got.dims == ('index', 'level') and got.shape == (n, 2)
    for k in range(2):
      want = sci.interpn((lat, lon), values[:, k].astype(np.float64), np.stack([slat, slon], axis=-1), method=method,
                         bounds_error=False, fill_value=None if extrapolate else np.nan)
      np.testing.assert_allclose(got.values[:, k], want, rtol=2e-6, atol=2e-6, equal_nan=True)
    np.testing.assert_array_equal(got.coords['elevation'].values, ref.coords['elevation'].values)   # the targets' coordinates come along
    np.testing.assert_array_equal(got.coords['latitude'].values, slat)
  torch = pytest.importorskip('torch')
  dt = xr.DataArray(torch.from_numpy(values), dims=da.dims, coords={'latitude': lat, 'level': [500, 850], 'longitude': lon})
  on_t = interpolations.interp(dt, {'latitude': ref['latitude'], 'longitude': ref['longitude']}, method, True)
  assert xr._is_torch(on_t.data)  # pylint: disable=protected-access
  np.testing.assert_allclose(np.asarray(on_t.values), interpolations.interp(da, {'latitude': ref['latitude'], 'longitude': ref['longitude']},
                                                                            method, True).values, rtol=1e-5, atol=1e-5)
  grid_t = interpolations.interp(dt, {'latitude': np.array([-10.0, 33.0]), 'longitude': np.array([5.0, 100.0, 200.0])}, method, True)
  grid_n = interpolations.interp(da, {'latitude': np.array([-10.0, 33.0]), 'longitude': np.array([5.0, 100.0, 200.0])}, method, True)
  assert xr._is_torch(grid_t.data) and grid_t.dims == grid_n.dims  # pylint: disable=protected-access
  np.testing.assert_allclose(np.asarray(grid_t.values), grid_n.values, rtol=1e-5, atol=1e-5)


def test_interp_errors_and_time_axes():
  da = xr.DataArray(np.arange(4.0), dims=('time',), coords={'time': np.datetime64('2020-01-01', 'ns') + np.arange(4) * np.timedelta64(6, 'h')})
  at = interpolations.interp(da, {'time': np.array(['2020-01-01T03', '2020-01-01T15'], dtype='datetime64[ns]')})
  np.testing.assert_allclose(at.values, [0.5, 2.5])
  with pytest.raises(ValueError, match='not in'):
    interpolations.interp(da, {'latitude': np.array([0.0])})
  with pytest.raises(ValueError, match='unsupported interpolation method'):
    interpolations.interp(da, {'time': da['time'].values}, method='cubic')
  twice = xr.DataArray(np.arange(3.0), dims=('x',), coords={'x': [0.0, 1.0, 1.0]})
  with pytest.raises(ValueError, match='repeated values'):
    interpolations.interp(twice, {'x': np.array([0.5])})


# ---- the reference's cases (interpolations_test.py:24-190) ---------------------------------------------------------------------------
def test_interpolate_to_reference_coords():
  reference, predictions = _grid(10), _grid(25)                          # fields of zeros: the frames are what is compared
  interpolation = interpolations.InterpolateToReferenceCoords(method='linear', dims=['latitude', 'longitude'], wrap_longitude=True)
  out = interpolation.interpolate(predictions, reference)
  for name in reference:
    assert out[name].dims == reference[name].dims and out[name].shape == reference[name].shape
    np.testing.assert_array_equal(np.asarray(out[name].values), np.asarray(reference[name].values))
    for d in ('latitude', 'longitude'):
      np.testing.assert_array_equal(out[name][d].values, reference[name][d].values)


def test_interpolate_to_fixed_coords_and_back():
  predictions = _grid(25, seed=3)
  coords = {'latitude': np.arange(-90, 90, 10), 'longitude': np.arange(0, 360, 10)}
  fixed = interpolations.InterpolateToFixedCoords(method='linear', coords=coords, wrap_longitude=True)
  out = fixed.interpolate(predictions)
  np.testing.assert_array_equal(out['2m_temperature']['latitude'].values, coords['latitude'])
  np.testing.assert_array_equal(out['2m_temperature']['longitude'].values, coords['longitude'])
  # cyclic in longitude: 355 E lies between the last meridian (337.5) and the first (0 = 360)
  src = predictions['2m_temperature']
  lon = src['longitude'].values
  wrapped = interpolations.InterpolateToFixedCoords('linear', {'longitude': np.array([355.0])}, wrap_longitude=True).interpolate_data_array(src)
  w = (355.0 - lon[-1]) / (360.0 - lon[-1])
  np.testing.assert_allclose(np.asarray(wrapped.isel(longitude=0).values),
                             np.asarray(src.isel(longitude=-1).values) * (1 - w) + np.asarray(src.isel(longitude=0).values) * w, rtol=1e-12)
  both = interpolations.MultipleInterpolation([fixed, interpolations.InterpolateToReferenceCoords(
      method='linear', dims=['latitude', 'longitude'], wrap_longitude=True)])
  back = both.interpolate(predictions, reference=predictions)             # interpolations_test.py:91-130: back on the original grid
  np.testing.assert_allclose(back['2m_temperature']['latitude'].values, src['latitude'].values)
  np.testing.assert_allclose(back['2m_temperature']['longitude'].values, lon)


def test_neighborhood_threshold_probabilities():
  predictions = _grid(15, seed=4)
  interpolation = interpolations.NeighborhoodThresholdProbabilities(neighborhood_sizes=[1, 3, 5], thresholds=[0.1, 0.9], wrap_longitude=True)
  out = interpolation.interpolate(predictions)['2m_temperature']
  assert {'smoothing_neighborhood', 'threshold_value'} <= set(out.dims) and out.sizes['smoothing_neighborhood'] == 3
  values = np.asarray(out.values, dtype=np.float64)
  assert values.max() <= 1.0 and values.min() >= 0.0
  src = np.asarray(predictions['2m_temperature'].values)
  n1 = out.sel(smoothing_neighborhood=1, threshold_value=0.9)
  np.testing.assert_array_equal(np.asarray(n1.transpose(*predictions['2m_temperature'].dims).values), (src > 0.9).astype(np.float64))


def test_interpolate_to_reference_coords_empty_reference():
  gridded = xr.DataArray(np.ones((2, 10, 20)), dims=['sample', 'latitude', 'longitude'], name='t2m',
                         coords={'sample': [1, 2], 'latitude': np.arange(10), 'longitude': np.arange(20)})
  empty = xr.DataArray(np.zeros((0,)), dims=['index'], name='t2m',
                       coords={'latitude': ('index', np.zeros((0,))), 'longitude': ('index', np.zeros((0,))), 'index': np.zeros((0,))})
  out = interpolations.InterpolateToReferenceCoords(method='linear', dims=['latitude', 'longitude']).interpolate_data_array(gridded, empty)
  assert out.dims == ('sample', 'index') and out.sizes['sample'] == 2 and out.sizes['index'] == 0
  np.testing.assert_array_equal(out['sample'].values, [1, 2])


def test_crop_to_box():
  rng = np.random.default_rng(5)
  lats, lons = np.arange(-85, 86, 10), np.arange(0, 359, 18)
  da = xr.DataArray(rng.random((len(lats), len(lons))), dims=['latitude', 'longitude'], name='t2m', coords={'latitude': lats, 'longitude': lons})
  cropped = interpolations.CropToBox(lat_min=-30, lat_max=30, lon_min=60, lon_max=180).interpolate_data_array(da)
  np.testing.assert_array_equal(cropped['latitude'].values, [-25, -15, -5, 5, 15, 25])
  np.testing.assert_array_equal(cropped['longitude'].values, np.arange(72, 181, 18))
  np.testing.assert_array_equal(cropped.values, da.values[6:12, 4:11])
  flipped = interpolations.CropToBox(-30, 30, 60, 180).interpolate_data_array(da.isel(latitude=np.arange(len(lats))[::-1]))
  np.testing.assert_array_equal(flipped.values, cropped.values)         # sorted ascending first
  with pytest.raises(ValueError, match='Invalid longitudes.*'):
    interpolations.CropToBox(lat_min=-90, lat_max=90, lon_min=300, lon_max=60)
  with pytest.raises(ValueError, match='Invalid latitudes.*'):
    interpolations.CropToBox(lat_min=10, lat_max=-10, lon_min=0, lon_max=10)


def test_subsample():
  rng = np.random.default_rng(6)
  lats, lons = np.arange(0, 100, 1.0), np.arange(0, 200, 1.0)
  da = xr.DataArray(rng.random((100, 200)), dims=['latitude', 'longitude'], name='t2m', coords={'latitude': lats, 'longitude': lons})
  out = interpolations.Subsample(dims=['latitude', 'longitude'], stride=10).interpolate_data_array(da)
  assert out.shape == (10, 20)
  np.testing.assert_array_equal(out['latitude'].values, lats[::10])
  np.testing.assert_array_equal(out.values, da.values[::10, ::10])
  same = interpolations.Subsample(dims=['latitude', 'longitude'], stride=1).interpolate_data_array(da)
  assert same.equals(da)
  only_lat = xr.DataArray(rng.random(10), dims=['latitude'], coords={'latitude': lats[:10]})
  out = interpolations.Subsample(dims=['latitude', 'longitude'], stride=2).interpolate_data_array(only_lat)
  assert out.sizes == {'latitude': 5}
  out = interpolations.Subsample(dims=['latitude'], stride=3).interpolate_data_array(da.isel(latitude=slice(0, 12), longitude=slice(0, 20)))
  assert out.shape == (4, 20)
  with pytest.raises(ValueError, match='stride must be >= 1'):
    interpolations.Subsample(dims=['latitude'], stride=0)
  out = interpolations.Subsample(dims=['latitude', 'longitude'], stride=2).interpolate({'t2m': da})
  assert out['t2m'].shape == (50, 100)


# ---- stations ---------------------------------------------------------------------------------------------------------------------------
def test_grid_to_sparse_with_altitude_adjustment():
  lat, lon = np.linspace(-10, 10, 5), np.linspace(0, 40, 9)
  orography = xr.DataArray(np.full((5, 9), 500.0), dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  station_height = np.array([500.0, 1500.0, 550.0, 900.0, 1900.0, 5000.0])
  n = station_height.size
  ref = xr.DataArray(np.zeros(n), dims=('index',), coords={
      'index': np.arange(n), 'latitude': (('index',), np.linspace(-8, 8, n)), 'longitude': (('index',), np.linspace(3, 37, n)),
      'elevation': (('index',), station_height)})
  adjust = interpolations.GridToSparseWithAltitudeAdjustment(method='linear', grid_elevation=orography)
  make = lambda name, value: xr.DataArray(np.full((2, 5, 9), value), dims=('time', 'latitude', 'longitude'), name=name,
                                          coords={'time': [0, 1], 'latitude': lat, 'longitude': lon})
  t2m = adjust.interpolate_data_array(make('2m_temperature', 280.0), ref)
  assert t2m.dims == ('time', 'index')
  dz = station_height - 500.0
  dz[np.abs(dz) >= 1500] = 0                                            # an implausible difference is not applied
  np.testing.assert_allclose(t2m.values, np.broadcast_to(280.0 - 0.0065 * dz, (2, n)), rtol=1e-12)
  wind = adjust.interpolate_data_array(make('10m_wind_speed', 10.0), ref)
  factor = np.where(dz < 100, 1.0, np.where(dz < 1100, 1 + 0.002 * (dz - 100), 3.0))
  np.testing.assert_allclose(wind.values, np.broadcast_to(10.0 * factor, (2, n)), rtol=1e-12)
  other = adjust.interpolate_data_array(make('total_precipitation', 1.0), ref)
  np.testing.assert_allclose(other.values, 1.0)
  assert 'grid_elevation' not in other.coords and 'grid_elevation' in t2m.coords
  clipped = interpolations.InterpolateToReferenceCoords('linear', clip_reference_coords=['latitude']).interpolate_data_array(
      make('x', 1.0).isel(latitude=slice(1, 4)), ref)
  assert clipped.sizes['index'] == int(((ref.coords['latitude'].values >= -5) & (ref.coords['latitude'].values <= 5)).sum())


def test_a_loader_that_regrids_its_chunks_to_the_targets():
  coarse = _grid(30, seed=7)
  fine_targets = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-04T00', time_resolution_hours=12,
                                            spatial_resolution_in_degrees=10, random=True, seed=8)
  lt = data_loaders.TargetsFromXarray(ds=fine_targets, variables=['2m_temperature'])
  lp = data_loaders.PredictionsFromXarray(ds=coarse, variables=['2m_temperature'], interpolation=interpolations.InterpolateToReferenceCoords(
      method='linear', dims=['latitude', 'longitude'], wrap_longitude=True))
  init_times = np.array(['2020-01-01T00', '2020-01-01T12'], dtype='datetime64[ns]')
  lead_times = np.array([0, 24], dtype='timedelta64[h]')
  t = lt.load_chunk(init_times, lead_times)
  p = lp.load_chunk(init_times, lead_times, reference=t)
  a, b = p['2m_temperature'], t['2m_temperature']
  assert a.sizes == b.sizes
  for d in ('latitude', 'longitude', 'init_time', 'lead_time'):
    np.testing.assert_array_equal(a[d].values, b[d].values)
  src = coarse['2m_temperature']
  lat30 = src['latitude'].values
  i = int(np.nonzero(lat30 == 30.0)[0][0])
  same_lat = a.sel(latitude=30.0, longitude=40.0)                        # a third of the way from the 30 E to the 60 E column of the coarse grid
  want = src.isel(latitude=i).sel(longitude=30.0) * (2 / 3) + src.isel(latitude=i).sel(longitude=60.0) * (1 / 3)
  got = same_lat.transpose('lead_time', 'init_time').values
  np.testing.assert_allclose(got, want.sel(prediction_timedelta=lead_times.astype('timedelta64[ns]'), time=init_times).values, rtol=1e-12)


@pytest.mark.parametrize('seed', range(12))
def test_random_interpolations_against_scipy(seed):
  """Random frames: 2 to 4 dims in random order, ascending or descending source axes, one to three of them interpolated --
  orthogonally to random targets, or pointwise to random points -- in both methods, against scipy.interpolate.interpn."""
  rng = np.random.default_rng(100 + seed)
  ndim = int(rng.integers(2, 5))
  names = ['a', 'b', 'c', 'd'][:ndim]
  sizes = [int(rng.integers(2, 7)) for _ in names]
  axes = {}
  for n, s in zip(names, sizes):
    ax = np.sort(rng.uniform(-10, 10, s))
    while np.diff(ax).min() < 1e-3:
      ax = np.sort(rng.uniform(-10, 10, s))
    axes[n] = ax[::-1].copy() if rng.random() < 0.4 else ax
  values = rng.normal(size=sizes)
  order = list(rng.permutation(ndim))
  da = xr.DataArray(np.transpose(values, order), dims=[names[i] for i in order], coords=axes)
  k = int(rng.integers(1, min(3, ndim) + 1))
  chosen = list(rng.choice(names, size=k, replace=False))
  method = 'linear' if rng.random() < 0.6 else 'nearest'
  extrapolate = bool(rng.random() < 0.5)
  fill = None if extrapolate else np.nan
  asc = {n: (axes[n], False) if axes[n][0] < axes[n][-1] else (axes[n][::-1], True) for n in names}

  def scipy_at(points_by_dim, others_index):
    """Values at points (arrays of one shape per chosen dim) for one index into the other dims."""
    grid = tuple(asc[n][0] for n in chosen)
    sub = values
    # bring chosen dims to the front in `chosen` order, fix the others
    perm = [names.index(n) for n in chosen] + [i for i, n in enumerate(names) if n not in chosen]
    sub = np.transpose(sub, perm)[(slice(None),) * len(chosen) + tuple(others_index)]
    for axis, n in enumerate(chosen):
      if asc[n][1]:
        sub = np.flip(sub, axis=axis)
    pts = np.stack([points_by_dim[n] for n in chosen], axis=-1)
    return sci.interpn(grid, sub, pts, method=method, bounds_error=False, fill_value=fill)

  others = [n for n in names if n not in chosen]
  if rng.random() < 0.5:                                                 # orthogonal
    targets = {n: rng.uniform(-12, 12, int(rng.integers(1, 5))) for n in chosen}
    got = interpolations.interp(da, targets, method, extrapolate).transpose(*chosen, *others)
    mesh = np.meshgrid(*[targets[n] for n in chosen], indexing='ij')
    for idx in np.ndindex(*[len(axes[n]) for n in others]):
      want = scipy_at(dict(zip(chosen, mesh)), idx)
      mine = np.asarray(got.values)[(slice(None),) * len(chosen) + idx]
      _assert_close_away_from_ties(mine, want, method)
  else:                                                                  # pointwise
    npts = int(rng.integers(1, 9))
    pts = {n: rng.uniform(-12, 12, npts) for n in chosen}
    targets = {n: xr.DataArray(v, dims=('points',)) for n, v in pts.items()}
    got = interpolations.interp(da, targets, method, extrapolate).transpose('points', *others)
    for idx in np.ndindex(*[len(axes[n]) for n in others]):
      want = scipy_at(pts, idx)
      mine = np.asarray(got.values)[(slice(None),) + idx]
      _assert_close_away_from_ties(mine, want, method)


def _assert_close_away_from_ties(mine, want, method):
  if method == 'linear':
    np.testing.assert_allclose(mine, want, rtol=1e-9, atol=1e-9, equal_nan=True)
  else:                                                                  # (random targets do not sit on midpoints)
    np.testing.assert_allclose(mine, want, rtol=0, atol=0, equal_nan=True)
