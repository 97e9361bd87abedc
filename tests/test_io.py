"""Persistence round trips (weatherbenchX/beam_pipeline.py:402-443, weatherbenchX/beam_utils.py:64-101 counterpart)."""
import os

import numpy as np
import scipy.io

from weatherbenchx_amd import io as wio
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.aggregation import AggregationState


def _state():
  lead = (np.arange(3) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  init = np.array(['2020-01-01T00', '2020-01-02T12'], dtype='datetime64[ns]')
  coords = {'lead_time': lead, 'region': np.array(['global', 'northern-hemisphere']), 'init_time': init}
  rng = np.random.default_rng(0)
  a = xr.DataArray(rng.normal(size=(2, 3, 2)), dims=('init_time', 'lead_time', 'region'), coords=coords)
  b = xr.DataArray(rng.normal(size=(3,)), dims=('lead_time',), coords={'lead_time': lead})
  return AggregationState({'SquaredError': {'z': a, 't2m': b}, 'CRPSSkill_number': {'t2m': b * 2}},
                          {'SquaredError': {'z': a * 0 + 5, 't2m': b * 0 + 7}, 'CRPSSkill_number': {'t2m': b * 0 + 1}})


def test_aggregation_state_file_round_trip(tmp_path):
  st = _state()
  path = os.path.join(tmp_path, 'sub', 'state.nc')
  wio.write_aggregation_state(st, path)
  assert os.path.exists(path) and not [f for f in os.listdir(os.path.dirname(path)) if f.startswith('.tmp_')]
  back = wio.read_aggregation_state(path)
  for tree_a, tree_b in ((st.sum_weighted_statistics, back.sum_weighted_statistics), (st.sum_weights, back.sum_weights)):
    for stat in tree_a:
      for var in tree_a[stat]:
        x, y = tree_a[stat][var], tree_b[stat][var]
        assert x.dims == y.dims
        np.testing.assert_allclose(x.values, y.values, rtol=0, atol=0)
        for d in x.dims:
          np.testing.assert_array_equal(x[d].values, y[d].values)
  # the file is plain NetCDF-3 with the reference's naming
  f = scipy.io.netcdf_file(path, 'r', mmap=False)
  assert 'SquaredError#z#sum_weighted_statistics' in f.variables and 'SquaredError#z#sum_weights' in f.variables
  assert f.variables['init_time'].units == b'seconds since 1970-01-01 00:00:00'
  f.close()
  # states read back keep adding up
  np.testing.assert_allclose((back + back).sum_weights['SquaredError']['t2m'].values, 14.0)


def test_metrics_file_round_trip(tmp_path):
  vals = xr.Dataset({'rmse.z': xr.DataArray(np.array([[1.0, 2.0]]), dims=('level', 'region'),
                                            coords={'level': np.array([500]), 'region': np.array(['a', 'bb'])}),
                     'bias.t2m': xr.DataArray(np.float64(0.25))})
  path = os.path.join(tmp_path, 'metrics.nc')
  wio.write_metrics(vals, path)
  back = wio.open_dataset(path)
  assert set(back) == {'rmse.z', 'bias.t2m'}
  np.testing.assert_allclose(back['rmse.z'].values, [[1.0, 2.0]])
  assert list(back['rmse.z']['region'].values) == ['a', 'bb'] and back['rmse.z']['level'].values.tolist() == [500]
  assert float(back['bias.t2m'].values) == 0.25


def test_non_index_coordinates_round_trip(tmp_path):
  """Accumulators that keep init_time and lead_time carry the 2-D coordinate valid_time(init_time, lead_time) (targets loaders,
  xarray_loaders.py:259-262); station results carry latitude / longitude over `index`.  They are written as CF auxiliary
  coordinates (the `coordinates` attribute) and come back as coordinates, not as data variables."""
  st = _state()
  a = st.sum_weighted_statistics['SquaredError']['z']
  valid = a['init_time'].values[:, None] + a['lead_time'].values[None, :]
  for tree in (st.sum_weighted_statistics, st.sum_weights):
    tree['SquaredError']['z'] = tree['SquaredError']['z'].assign_coords(valid_time=(('init_time', 'lead_time'), valid))
  path = os.path.join(tmp_path, 'state.nc')
  wio.write_aggregation_state(st, path)
  back = wio.read_aggregation_state(path)
  assert set(back.sum_weighted_statistics) == set(st.sum_weighted_statistics)
  z = back.sum_weighted_statistics['SquaredError']['z']
  np.testing.assert_array_equal(z.coords['valid_time'].values, valid)
  assert z.coords['valid_time'].dims == ('init_time', 'lead_time')
  assert 'valid_time' not in back.sum_weighted_statistics['SquaredError']['t2m'].coords    # only where its dims are
  stations = xr.Dataset({'rmse.t': xr.DataArray(np.arange(3.0), dims=('index',), coords={
      'index': np.arange(3), 'latitude': (('index',), np.array([10.0, 20.0, 30.0])), 'stationName': (('index',), np.array(['A', 'BB', 'C']))})})
  wio.write_metrics(stations, os.path.join(tmp_path, 'm.nc'))
  m = wio.open_dataset(os.path.join(tmp_path, 'm.nc'))
  assert set(m) == {'rmse.t'} and m['rmse.t'].coords['latitude'].values.tolist() == [10.0, 20.0, 30.0]
  assert m['rmse.t'].coords['stationName'].values.tolist() == ['A', 'BB', 'C']
