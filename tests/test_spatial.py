"""Fractions skill score (weatherbenchx_amd/metrics/spatial.py): the reference's known answers (metrics/metrics_test.py:172-406)
restated; the sliding-window neighbourhood mean against the window written out in the oracle, which is pinned here against
scipy.ndimage.convolve1d (what the reference calls, spatial.py:44-45); host and tensor payloads; FSS through the Aggregator on
both backends."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import spatial


def _scipy_mean(x, n, wrap_longitude):
  """The reference's two convolve1d passes."""
  kernel = np.ones(n, dtype=np.float32) / n
  out = ndimage.convolve1d(ndimage.convolve1d(x.astype(np.float32), kernel, mode='wrap', axis=0), kernel, mode='wrap', axis=1)
  h = (n - 1) // 2
  out[:h] = 0
  out[-h:] = 0
  if not wrap_longitude:
    out[:, :h] = 0
    out[:, -h:] = 0
  return out


@pytest.mark.parametrize('n', [3, 5, 9])
@pytest.mark.parametrize('wrap', [False, True])
def test_neighbourhood_mean_against_the_window_and_scipy(n, wrap):
  rng = np.random.default_rng(n)
  x = (rng.random((4, 13, 17)) > 0.6).astype(np.float64)
  want = np.stack([O.neighborhood_mean(f, n, wrap) for f in x])
  np.testing.assert_allclose(want, np.stack([_scipy_mean(f, n, wrap) for f in x]), atol=2e-6)          # the oracle's pin
  got = spatial.convolve2d_wrap_longitude(x.copy(), n, wrap)
  assert got.dtype == np.float32
  np.testing.assert_allclose(got, want, atol=1e-6)
  torch = pytest.importorskip('torch')
  got_t = spatial.convolve2d_wrap_longitude(torch.from_numpy(x.copy()), n, wrap)
  assert got_t.dtype == torch.float32
  np.testing.assert_allclose(got_t.numpy(), want, atol=1e-6)


def test_neighbourhood_mean_corner_cases():
  x = np.ones((5, 5))
  x[0, 0] = np.nan
  out = spatial.convolve2d_wrap_longitude(x, 3)                         # metrics_test.py:215-231
  np.testing.assert_allclose(out, np.array([[0.0, 0.0, 0.0, 0.0, 0.0], [0.0, np.nan, 1.0, 1.0, 0.0], [0.0, 1.0, 1.0, 1.0, 0.0],
                                            [0.0, 1.0, 1.0, 1.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0]]))
  assert spatial.convolve2d_wrap_longitude(x, 1) is x                   # n = 1: the input itself
  with pytest.raises(ValueError, match='must be odd'):
    spatial.convolve2d_wrap_longitude(x, 4)
  # a window larger than the grid wraps around more than once; with wrap_longitude the columns are kept
  y = np.arange(12.0).reshape(4, 3)
  np.testing.assert_allclose(spatial.convolve2d_wrap_longitude(y.copy(), 5, True), O.neighborhood_mean(y, 5, True), atol=1e-5)
  assert spatial.get_suffix([1, 3], True) == '1,3_wrap_longitude' and spatial.get_suffix(5) == '5'


def test_fss():
  """metrics_test.py:172-262."""
  p = xr.DataArray(np.array([1, 0, 1, 0, 0, 1]), dims=['longitude'], name='precipitation').expand_dims(latitude=3)
  t = xr.DataArray(np.array([1, 0, 0, 1, 0, 1]), dims=['longitude'], name='precipitation').expand_dims(latitude=3)
  prediction, target = {'precipitation': p}, {'precipitation': t}
  metrics = {'fss_no_wrap': spatial.FSS(neighborhood_size_in_pixels=[1, 3], wrap_longitude=False),
             'fss_wrap': spatial.FSS(neighborhood_size_in_pixels=[1, 3], wrap_longitude=True)}
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, prediction, target)
  stats = xarray_tree.map_structure(lambda x: x.mean(['latitude', 'longitude']), stats)
  no_wrap = metrics_base.compute_metric_from_statistics(metrics['fss_no_wrap'], stats)['precipitation']
  wrap = metrics_base.compute_metric_from_statistics(metrics['fss_wrap'], stats)['precipitation']
  np.testing.assert_allclose(np.asarray(no_wrap.sel(neighborhood_size=1).values), 4 / 6)
  np.testing.assert_allclose(np.asarray(wrap.sel(neighborhood_size=1).values), 4 / 6)
  assert float(np.asarray(wrap.sel(neighborhood_size=3).values)) > float(np.asarray(no_wrap.sel(neighborhood_size=3).values))
  for w, got in ((False, no_wrap), (True, wrap)):
    np.testing.assert_allclose(float(np.asarray(got.sel(neighborhood_size=3).values)),
                               O.fractions_skill_score(np.asarray(p.values), np.asarray(t.values), 3, w), rtol=1e-6)
  # NaNs with n = 1: no masks anywhere, the squares of the inputs come back
  p1 = {'precipitation': xr.DataArray(np.array([[1, 0, np.nan, 1]]), dims=['latitude', 'longitude'])}
  t1 = {'precipitation': xr.DataArray(np.array([[0, np.nan, 1, 0]]), dims=['latitude', 'longitude'])}
  fss = spatial.FSS(neighborhood_size_in_pixels=1)
  stats1 = metrics_base.compute_unique_statistics_for_all_metrics({'fss': fss}, p1, t1)
  np.testing.assert_allclose(np.asarray(stats1[fss.statistics['SquaredPredictionFraction'].unique_name]['precipitation'].values),
                             [[1.0, 0.0, np.nan, 1.0]])
  np.testing.assert_allclose(np.asarray(stats1[fss.statistics['SquaredTargetFraction'].unique_name]['precipitation'].values),
                             [[0.0, np.nan, 1.0, 0.0]])
  assert fss.statistics['SquaredFractionsError'].unique_name == 'SquaredFractionsError_1'


def _masked_pair():
  pred = xr.DataArray(np.array([[1.0, 2.0], [3.0, 4.0]]), dims=['latitude', 'longitude'], name='precipitation',
                      coords={'mask': (('latitude', 'longitude'), np.array([[True, False], [True, True]]))})
  target = xr.DataArray(np.array([[5.0, 6.0], [7.0, 8.0]]), dims=['latitude', 'longitude'], name='precipitation',
                        coords={'mask': (('latitude', 'longitude'), np.array([[True, True], [False, True]]))})
  return pred, target


def test_get_fss_mask():
  """metrics_test.py:263-319."""
  pred, target = _masked_pair()
  np.testing.assert_array_equal(np.asarray(spatial.get_fss_mask(pred, target, 1, combine_mask=True).values), [[True, False], [False, True]])
  np.testing.assert_array_equal(np.asarray(spatial.get_fss_mask(pred, target, 1, combine_mask=False).values), [[True, True], [False, True]])
  bare_t, bare_p = target.drop_vars('mask'), pred.drop_vars('mask')
  np.testing.assert_array_equal(np.asarray(spatial.get_fss_mask(pred, bare_t, 1, combine_mask=False).values), [[True, False], [True, True]])
  assert spatial.get_fss_mask(bare_p, bare_t, 1, combine_mask=False) is None
  ones = xr.DataArray(np.ones((5, 5)), dims=['latitude', 'longitude'], coords={'mask': (('latitude', 'longitude'), np.ones((5, 5), dtype=bool))})
  inner = np.zeros((5, 5), dtype=bool)
  inner[1:-1, 1:-1] = True
  np.testing.assert_array_equal(np.asarray(spatial.get_fss_mask(ones, ones, 3, combine_mask=False).values), inner)
  both = spatial.get_fss_mask(ones, ones, [1, 3])
  assert both.dims == ('neighborhood_size', 'latitude', 'longitude') and np.asarray(both.values)[0].all()
  np.testing.assert_array_equal(np.asarray(both.values)[1], inner)


@pytest.mark.parametrize('cls', [spatial.SquaredPredictionFraction, spatial.SquaredTargetFraction, spatial.SquaredFractionsError])
def test_fss_statistics_mask_propagation(cls):
  """metrics_test.py:321-406."""
  pred, target = _masked_pair()
  combined = cls(neighborhood_size_in_pixels=1, combine_mask=True)._compute_per_variable(pred, target)
  np.testing.assert_array_equal(np.asarray(combined.coords['mask'].values), [[True, False], [False, True]])
  targets_only = cls(neighborhood_size_in_pixels=1, combine_mask=False)._compute_per_variable(pred, target)
  np.testing.assert_array_equal(np.asarray(targets_only.coords['mask'].values), [[True, True], [False, True]])
  want = {'SquaredPredictionFraction': pred.values ** 2, 'SquaredTargetFraction': target.values ** 2,
          'SquaredFractionsError': (pred.values - target.values) ** 2}[cls.__name__]
  np.testing.assert_allclose(np.asarray(combined.values), want)


def test_fss_of_fields_through_the_aggregator(backend):
  del backend
  rng = np.random.default_rng(43)
  base_field = rng.random((3, 12, 16))
  p = (base_field + 0.25 * rng.normal(size=base_field.shape) > 0.7).astype(np.float32)
  t = (base_field > 0.7).astype(np.float32)
  cs = {'time': np.arange(3), 'latitude': np.linspace(-55, 55, 12), 'longitude': np.arange(16) * 22.5}
  dims = ('time', 'latitude', 'longitude')
  pred, targ = {'rain': xr.DataArray(p, dims=dims, coords=cs)}, {'rain': xr.DataArray(t, dims=dims, coords=cs)}
  metrics = {'fss': spatial.FSS([1, 3, 5], wrap_longitude=True)}
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, pred, targ)
  out = aggregation.Aggregator(reduce_dims=list(dims)).aggregate_statistics(stats).metric_values(metrics)['fss.rain']
  assert out.dims == ('neighborhood_size',)
  for n in (1, 3, 5):
    np.testing.assert_allclose(float(np.asarray(out.sel(neighborhood_size=n).values)), O.fractions_skill_score(p, t, n, True), rtol=2e-6)
  values = np.asarray(out.values)
  assert values[0] < values[1] < values[2] < 1                           # skill grows with the neighbourhood
  # with a NaN mask on the targets: masked aggregation over the pixels whose whole neighbourhood is valid
  valid = np.ones(t.shape, dtype=bool)
  valid[1, 4:7, 5:9] = False
  t_nan = np.where(valid, t, np.nan)
  targ_m = {'rain': xr.DataArray(t_nan, dims=dims, coords=dict(cs, mask=(dims, valid)))}
  stats = metrics_base.compute_unique_statistics_for_all_metrics({'fss': spatial.FSS(3, wrap_longitude=True)}, pred, targ_m)
  out = aggregation.Aggregator(reduce_dims=list(dims), masked=True).aggregate_statistics(stats).metric_values(
      {'fss': spatial.FSS(3, wrap_longitude=True)})['fss.rain']
  pf = np.stack([O.neighborhood_mean(f, 3, True) for f in p])
  tf = np.stack([O.neighborhood_mean(f, 3, True) for f in t_nan])
  ok = np.isclose(np.stack([O.neighborhood_mean(f, 3, True) for f in valid.astype(float)]), 1.0)
  mean = lambda x: np.where(ok, x, 0).sum() / ok.sum()
  np.testing.assert_allclose(float(np.asarray(out.values)), 1 - mean((pf - tf) ** 2) / (mean(pf ** 2) + mean(tf ** 2)), rtol=2e-6)


def test_tensor_payloads_stay_tensors():
  torch = pytest.importorskip('torch')
  rng = np.random.default_rng(47)
  p, t = (rng.random((2, 9, 11)) > 0.5).astype(np.float32), (rng.random((2, 9, 11)) > 0.5).astype(np.float32)
  dims = ('time', 'latitude', 'longitude')
  stat = spatial.SquaredFractionsError([1, 3])
  a = stat.compute({'v': xr.DataArray(p, dims=dims)}, {'v': xr.DataArray(t, dims=dims)})['v']
  b = stat.compute({'v': xr.DataArray(torch.from_numpy(p), dims=dims)}, {'v': xr.DataArray(torch.from_numpy(t), dims=dims)})['v']
  assert xr._is_torch(b.data) and a.dims == b.dims == ('neighborhood_size',) + dims  # pylint: disable=protected-access
  np.testing.assert_allclose(np.asarray(b.values), np.asarray(a.values), atol=1e-6)


@pytest.mark.parametrize('seed', range(6))
def test_neighbourhood_mean_with_scattered_nans(seed):
  """NaNs poison exactly the windows that hold them (running sums would smear them over everything behind: they are counted
  separately), for every size and both boundary rules, also when windows wrap more than once."""
  rng = np.random.default_rng(200 + seed)
  nlat, nlon = int(rng.integers(4, 12)), int(rng.integers(4, 14))
  x = rng.normal(size=(2, nlat, nlon))
  x[rng.random(x.shape) < 0.08] = np.nan
  n = int(rng.choice([3, 5, 7]))
  wrap = bool(rng.random() < 0.5)
  want = np.stack([O.neighborhood_mean(f, n, wrap) for f in x])
  got = spatial.convolve2d_wrap_longitude(x.copy(), n, wrap)
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
  np.testing.assert_allclose(got, want, atol=2e-6, equal_nan=True)
