"""StationDensityWeighting (weatherbenchX/weighting.py:133-330): the reference's own known answers (weighting_test.py:48-232) and a
station-weighted RMSE through the Aggregator against the float64 oracle."""
import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic


def _sparse(lats, lons, values=None):
  n = len(lats)
  return xr.DataArray(np.ones(n) if values is None else np.asarray(values, float), dims=['index'],
                      coords={'latitude': (('index',), np.array(lats, float)), 'longitude': (('index',), np.array(lons, float))})


def test_haversine():
  z = np.array([0.0])
  assert float(weighting._haversine(z, z, z, z)[0]) == pytest.approx(0.0)  # pylint: disable=protected-access
  assert float(weighting._haversine(z, z, z, np.array([np.pi / 2]))[0]) == pytest.approx(np.pi / 2)  # pylint: disable=protected-access
  assert float(weighting._haversine(np.array([np.pi / 2]), z, np.array([-np.pi / 2]), z)[0]) == pytest.approx(np.pi)  # pylint: disable=protected-access


def test_station_density_known_answers():
  w = weighting.StationDensityWeighting().weights(_sparse([0, 0, 10, 10], [0, 10, 0, 10]))
  np.testing.assert_allclose(np.asarray(w.values), 1.0, atol=0.01)                      # well separated: equal weights
  w = np.asarray(weighting.StationDensityWeighting().weights(_sparse([0.0, 0.01, 0.02, 10.0], [0.0, 0.01, 0.02, 10.0])).values)
  assert w[3] > w[0] and w[3] > w[1] and w[3] > w[2]                                  # the isolated station weighs most
  w = weighting.StationDensityWeighting(return_normalized=True).weights(_sparse([0, 0.5, 1.0, 50.0], [0, 0.5, 1.0, 50.0]))
  assert float(np.asarray(w.values).mean()) == pytest.approx(1.0, abs=1e-5)
  w = np.asarray(weighting.StationDensityWeighting(return_normalized=False).weights(_sparse([0, 0, 10], [0, 0, 10])).values)
  np.testing.assert_allclose(w, [0.5, 0.5, 1.0], atol=1e-6)                             # two co-located stations: density 2
  assert float(np.asarray(weighting.StationDensityWeighting().weights(_sparse([45.0], [10.0])).values)[0]) == pytest.approx(1.0)
  s3 = _sparse([0, 1.0, 50.0], [0, 1.0, 50.0])
  small = np.asarray(weighting.StationDensityWeighting(alpha_0_degrees=0.5).weights(s3).values)
  large = np.asarray(weighting.StationDensityWeighting(alpha_0_degrees=2.0).weights(s3).values)
  assert small[2] / small[0] < large[2] / large[0]
  s4 = _sparse([0, 0.01, 0.02, 50.0], [0, 0.01, 0.02, 50.0])
  assert float(np.asarray(weighting.StationDensityWeighting().weights(s4).values).max()) > 1.5
  assert float(np.asarray(weighting.StationDensityWeighting(max_weight=1.5).weights(s4).values).max()) == pytest.approx(1.5)
  np.testing.assert_array_equal(np.asarray(weighting.StationDensityWeighting().weights(s3).values),
                                np.asarray(weighting.StationDensityWeighting(max_weight=None).weights(s3).values))
  multi = weighting.StationDensityWeighting(alpha_0_degrees=[0.5, 2.0]).weights(s3)
  assert multi.dims == ('index', 'weighting_alpha_0')
  np.testing.assert_array_equal(np.asarray(multi.coords['weighting_alpha_0'].values), [0.5, 2.0])
  np.testing.assert_allclose(np.asarray(multi.sel(weighting_alpha_0=0.5).values), small)
  np.testing.assert_allclose(np.asarray(multi.sel(weighting_alpha_0=2.0).values), large)
  with pytest.raises(ValueError, match='scalar or 1D'):
    weighting.StationDensityWeighting(alpha_0_degrees=[[1.0]]).weights(s3)


def test_station_density_is_the_identity_off_sparse_data():
  grid = xr.DataArray(np.ones((3, 4)), dims=['latitude', 'longitude'], coords={'latitude': np.arange(3.), 'longitude': np.arange(4.)})
  nocoords = xr.DataArray(np.array([1.0, 2.0]), dims=['station'])
  two_d = xr.DataArray(np.ones((2, 2)), dims=['x', 'y'], coords={'latitude': (('x', 'y'), np.arange(4.).reshape(2, 2)),
                                                                 'longitude': (('x', 'y'), np.arange(4.).reshape(2, 2))})
  for stat in (grid, nocoords, two_d):
    for alpha in (0.75, [0.5, 1.0]):
      w = weighting.StationDensityWeighting(alpha_0_degrees=alpha).weights(stat)
      assert float(np.asarray(w.values)) == 1.0 and 'weighting_alpha_0' not in w.dims


def test_blocked_angles_equal_the_dense_formula():
  rng = np.random.default_rng(0)
  lat, lon = rng.uniform(-80, 80, 700), rng.uniform(0, 360, 700)
  w = np.asarray(weighting.StationDensityWeighting(alpha_0_degrees=1.5).weights(_sparse(lat, lon)).values)
  la, lo = np.deg2rad(lat), np.deg2rad(lon)
  ang = weighting._haversine(la[:, None], lo[:, None], la[None, :], lo[None, :])  # pylint: disable=protected-access
  want = 1.0 / np.exp(-(ang / np.deg2rad(1.5)) ** 2).sum(axis=1)
  np.testing.assert_allclose(w, want / want.mean(), rtol=1e-12)


def test_station_weighted_rmse_through_the_aggregator(backend):
  rng = np.random.default_rng(1)
  n = 400
  lat, lon = rng.uniform(-60, 60, n), rng.uniform(0, 360, n)
  lat[:50], lon[:50] = 10 + rng.normal(size=50) * 0.05, 20 + rng.normal(size=50) * 0.05  # a dense cluster
  pv, tv = rng.normal(size=n), rng.normal(size=n)
  cs = {'index': np.arange(n), 'latitude': (('index',), lat), 'longitude': (('index',), lon)}
  p, t = {'v': xr.DataArray(pv, dims=('index',), coords=cs)}, {'v': xr.DataArray(tv, dims=('index',), coords=cs)}
  metrics = {'rmse': deterministic.RMSE()}
  agg = aggregation.Aggregator(reduce_dims=['index'], weigh_by=[weighting.StationDensityWeighting(alpha_0_degrees=1.0)])
  got = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t)).metric_values(metrics)['rmse.v']
  w = np.asarray(weighting.StationDensityWeighting(alpha_0_degrees=1.0).weights(p['v']).values)
  sws, sw, _ = O.aggregate((pv - tv) ** 2, ('index',), ['index'], weights=[(w, ('index',))])
  np.testing.assert_allclose(float(np.asarray(got.values)), np.sqrt(sws / sw), rtol=1e-9)
