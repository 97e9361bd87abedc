"""Statistic / Metric behaviour, restating weatherbenchX/metrics/metrics_test.py:44-98, 501-544, 603-660,
947-1006, 1276-1308 and metrics/base_test.py:24-91 against the drop-in classes, plus parity of every fused
lane against the float64 oracle on random data in several memory layouts (tolerance: rtol 1e-6 as north_star
states; the fp64 accumulators actually land at ~1e-12)."""
import itertools

import numpy as np
import pytest

import mock_data
from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import probabilistic
from weatherbenchx_amd.metrics import wrappers

RTOL = 1e-6


def compute_all_metrics(metrics, predictions, targets, reduce_dims, **kw):
  """metrics/metrics_test_utils.py:86-95."""
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, predictions, targets)
  return aggregation.Aggregator(reduce_dims=reduce_dims, **kw).aggregate_statistics(stats).metric_values(metrics)


# ---- protocol (base_test.py) -------------------------------------------------------------------------
def test_per_variable_statistics_on_subsets_of_variables(backend):
  class Stat1(metrics_base.PerVariableStatistic):
    def _compute_per_variable(self, predictions, targets):
      return None if targets.name == 'a' else predictions

  class Stat2(metrics_base.PerVariableStatistic):
    def _compute_per_variable(self, predictions, targets):
      return None if targets.name == 'b' else targets

  class Metric(metrics_base.PerVariableMetric):
    statistics = {'stat1': Stat1(), 'stat2': Stat2()}

    def _values_from_mean_statistics_per_variable(self, stats):
      return stats['stat1'] + stats['stat2']

  predictions = xr.Dataset(dict(a=xr.DataArray(1.0), b=xr.DataArray(2.0), c=xr.DataArray(3.0)))
  targets = xr.Dataset(dict(a=xr.DataArray(10.0), b=xr.DataArray(20.0), c=xr.DataArray(30.0)))
  result = compute_all_metrics({'metric': Metric(), 'stat1': Stat1(), 'stat2': Stat2()}, predictions, targets, [])
  assert 'stat1.a' not in result and 'stat1.b' in result and 'stat1.c' in result
  assert 'stat2.a' in result and 'stat2.b' not in result and 'stat2.c' in result
  assert set(k for k in result if k.startswith('metric.')) == {'metric.c'}
  np.testing.assert_allclose(result['metric.c'].values, 33.0)


def test_statistic_only_for_common_variables_and_plain_dict(backend):
  class Stat(metrics_base.PerVariableStatistic):
    def _compute_per_variable(self, predictions, targets):
      return abs(predictions - targets)

  predictions = xr.Dataset(dict(a=xr.DataArray(1.0), b=xr.DataArray(2.0)))
  targets = dict(b=xr.DataArray(20.0), c=xr.DataArray(30.0))
  result = Stat().compute(predictions, targets)
  assert isinstance(result, dict) and list(result) == ['b']


def test_failed_statistic_is_wrapped_in_value_error(backend):
  class Boom(metrics_base.PerVariableStatistic):
    def _compute_per_variable(self, predictions, targets):
      raise RuntimeError('boom')

  with pytest.raises(ValueError, match='Failed to compute statistic Boom'):
    metrics_base.compute_unique_statistics_for_all_metrics({'b': Boom()}, {'a': xr.DataArray(1.0)},
                                                           {'a': xr.DataArray(1.0)})


def test_statistics_are_deduplicated_by_unique_name(backend):
  metrics = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE()}
  p = {'v': xr.DataArray(np.ones((3, 4), np.float32), dims=('latitude', 'longitude'))}
  t = {'v': xr.DataArray(np.zeros((3, 4), np.float32), dims=('latitude', 'longitude'))}
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t)
  assert list(stats) == ['SquaredError']


# ---- deterministic known answers ---------------------------------------------------------------------
@pytest.mark.parametrize('split_variables', [False, True])
def test_statistics_computation(backend, split_variables):
  target = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-20T00',
                                          variables_2d=['2m_temperature', '10m_wind_speed'])
  prediction = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00',
                                              variables_2d=['2m_temperature', '10m_wind_speed']) + 1
  if split_variables:
    target, prediction = dict(target), dict(prediction)
  metrics = {'rmse': deterministic.RMSE()}
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, prediction, target)
  assert set(stats['SquaredError']) == set(target)
  assert stats['SquaredError']['2m_temperature'].mean() == 1.0      # materialised by the map kernel
  assert stats['SquaredError']['geopotential'].shape == prediction['geopotential'].shape
  for v in stats['SquaredError']:
    assert metrics_base.compute_metric_from_statistics(metrics['rmse'], stats)[v].dims == stats['SquaredError'][v].dims


def test_wind_vector_rmse(backend):
  names2d = ['10m_u_component_of_wind', '10m_v_component_of_wind']
  names3d = ['u_component_of_wind', 'v_component_of_wind']
  target = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-20T00', variables_2d=names2d,
                                          variables_3d=names3d) + 1
  prediction = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00',
                                              variables_2d=names2d, variables_3d=names3d)
  metrics = {'vector_rmse': deterministic.WindVectorRMSE(['u_component_of_wind', '10m_u_component_of_wind'],
                                                         ['v_component_of_wind', '10m_v_component_of_wind'],
                                                         ['wind', '10m_wind'])}
  results = compute_all_metrics(metrics, prediction, target, ['time', 'latitude', 'longitude'])
  assert set(results) == {'vector_rmse.wind', 'vector_rmse.10m_wind'}
  np.testing.assert_allclose(results['vector_rmse.wind'].values, np.sqrt(2))
  np.testing.assert_allclose(results['vector_rmse.10m_wind'].values, np.sqrt(2))
  assert set(results['vector_rmse.wind'].dims) == {'prediction_timedelta', 'level'}


def test_acc_is_one(backend):
  prediction = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-02T00').rename(
      time='init_time', prediction_timedelta='lead_time')
  target = prediction.copy()
  climatology = target.isel(init_time=0, lead_time=0, drop=True).expand_dims(
      dayofyear=np.arange(1, 367), hour=np.array([0, 6, 12, 18])) - 1
  metrics = {'acc': deterministic.ACC(climatology=climatology),
             'activity': deterministic.PredictionActivity(climatology=climatology)}
  results = compute_all_metrics(metrics, prediction, target, ['latitude', 'longitude'])
  for v in ('acc.2m_temperature', 'acc.geopotential', 'activity.geopotential'):
    np.testing.assert_allclose(results[v].values, 1.0)


# ---- ensemble known answers ----------------------------------------------------------------------------
@pytest.mark.parametrize('m,use_sort,fair', list(itertools.product([4, 5], [False, True], [True, False])))
def test_crps_equals_brute_force(backend, m, use_sort, fair):
  targets = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True, seed=10)
  predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True,
                                               ensemble_size=m, seed=11)
  metrics = {'crps': probabilistic.CRPSEnsemble(ensemble_dim='realization', use_sort=use_sort, fair=fair)}
  results = compute_all_metrics(metrics, predictions, targets, ['latitude', 'longitude'])
  for v in ['2m_temperature', 'geopotential']:
    p, t = predictions[v], targets[v]
    spread = abs(p - p.rename(realization='dummy')).mean(('latitude', 'longitude', 'realization', 'dummy'),
                                                         skipna=False) * (m / (m - int(fair)))
    skill = abs(t - p).mean(('latitude', 'longitude', 'realization'), skipna=False)
    xr.assert_allclose(skill - 0.5 * spread, results[f'crps.{v}'], rtol=1e-9, check_dim_order=False)


def test_crps_spread_needs_two_members(backend):
  p = {'v': xr.DataArray(np.zeros((1, 3)), dims=('number', 'x'))}
  t = {'v': xr.DataArray(np.zeros(3), dims=('x',))}
  with pytest.raises(ValueError, match='Failed to compute statistic CRPSSpread') as info:
    metrics_base.compute_unique_statistics_for_all_metrics({'c': probabilistic.CRPSEnsemble()}, p, t)
  assert 'n_ensemble < 2' in str(info.value.__cause__)
  with pytest.raises(ValueError, match='SpreadSkillRatio is no longer supported'):
    probabilistic.SpreadSkillRatio(ensemble_dim='number')


def test_spread_skill_ratio_near_one(backend):
  m = 5
  targets = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', variables_3d=[],
                                       random=True, seed=0)
  predictions = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', variables_3d=[],
                                           ensemble_size=m, random=True, seed=1)
  metrics = {'ssr': probabilistic.UnbiasedSpreadSkillRatio(ensemble_dim='realization'),
             'rmv': probabilistic.EnsembleRootMeanVariance(ensemble_dim='realization'),
             'urmse': probabilistic.UnbiasedEnsembleMeanRMSE(ensemble_dim='realization')}
  results = compute_all_metrics(metrics, predictions, targets, ['time', 'latitude', 'longitude'])
  n = np.prod(list(targets['2m_temperature'].shape))
  assert abs(float(results['ssr.2m_temperature'].values) - 1) < 4 / np.sqrt(n * m)
  np.testing.assert_allclose(results['rmv.2m_temperature'].values / results['urmse.2m_temperature'].values,
                             results['ssr.2m_temperature'].values)


def test_ensemble_averaged_metric_equals_reducing_over_members(backend):
  targets = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True, seed=3)
  predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True,
                                               ensemble_size=5, seed=4)
  expected = compute_all_metrics({'rmse': deterministic.RMSE()}, predictions, targets,
                                 ['latitude', 'longitude', 'realization'])
  actual = compute_all_metrics({'rmse': probabilistic.EnsembleAveragedMetric(deterministic.RMSE(),
                                                                             ensemble_dim='realization')},
                               predictions, targets, ['latitude', 'longitude'])
  for v in expected:
    xr.assert_allclose(actual[v], expected[v], rtol=1e-10, check_dim_order=False)


def test_mean_rmse_through_wrapped_metric(backend):
  rng = np.random.default_rng(5)
  p = {'v': xr.DataArray(rng.normal(size=(6, 7, 8)).astype(np.float32), dims=('number', 'latitude', 'longitude'))}
  t = {'v': xr.DataArray(rng.normal(size=(7, 8)).astype(np.float32), dims=('latitude', 'longitude'))}
  metrics = {'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')]),
             'crps': probabilistic.CRPSEnsemble(use_sort=True)}
  res = compute_all_metrics(metrics, p, t, ['latitude', 'longitude'])
  want = np.sqrt(((p['v'].values.astype(np.float64).mean(0) - t['v'].values) ** 2).mean())
  np.testing.assert_allclose(res['mean_rmse.v'].values, want, rtol=RTOL)
  stat = list(metrics['mean_rmse'].statistics.values())[0]
  assert stat.unique_name == "SquaredError_predictions_ensemble_mean_self._ensemble_dim='number'_self._skipna=False"


# ---- parity against the oracle in several memory layouts ----------------------------------------------
LAT = np.linspace(-87.1875, 87.1875, 32)
LON = np.arange(64) * 5.625
LAYOUTS = {
    'lon_fastest': ('init_time', 'lead_time', 'level', 'latitude', 'longitude'),
    'lat_fastest': ('init_time', 'lead_time', 'level', 'longitude', 'latitude'),   # real WeatherBench chunks
    'level_fastest': ('lead_time', 'init_time', 'latitude', 'longitude', 'level'),  # the reference's mock layout
}
SIZES = {'init_time': 2, 'lead_time': 3, 'level': 3, 'latitude': 32, 'longitude': 64}


def _field(rng, dims, dtype, offset=0.0):
  shape = [SIZES[d] for d in dims]
  coords = {'latitude': LAT, 'longitude': LON, 'level': np.array([500, 700, 850]),
            'init_time': np.array(['2020-01-01T00', '2020-01-02T00'], dtype='datetime64[ns]'),
            'lead_time': (np.arange(3) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')}
  return xr.DataArray((rng.normal(size=shape) + offset).astype(dtype), dims=dims, coords={d: coords[d] for d in dims})


REGIONS = {'global': ((-90, 90), (0, 360)), 'northern-hemisphere': ((20, 90), (0, 360)),
           'europe': ((35, 75), (-12.5, 42.5))}


@pytest.mark.parametrize('layout', list(LAYOUTS))
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('reduce_dims', [('init_time', 'latitude', 'longitude'), ('latitude', 'longitude'),
                                         ('init_time',), ('lead_time', 'level'), ()])
def test_deterministic_suite_matches_oracle(backend, layout, dtype, reduce_dims):
  rng = np.random.default_rng(0)
  dims = LAYOUTS[layout]
  p, t = _field(rng, dims, dtype, 280.0), _field(rng, dims, dtype, 280.0)
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=list(reduce_dims), weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS)])
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'z': p}, {'z': t})
  state = agg.aggregate_statistics(stats)
  w = (O.grid_area_weights(LAT), ('latitude',))
  _, masks = O.region_masks(LAT, LON, REGIONS)
  bins = ('region', masks, ('region', 'latitude', 'longitude'))
  for name, fn in (('Error', O.error), ('AbsoluteError', O.absolute_error), ('SquaredError', O.squared_error)):
    sws, sw, out_dims = O.aggregate(fn(p.values, t.values), dims, reduce_dims, weights=[w], bin_masks=[bins])
    got = state.sum_weighted_statistics[name]['z']
    assert set(got.dims) == set(out_dims)
    np.testing.assert_allclose(got.transpose(*out_dims).values, sws, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(state.sum_weights[name]['z'].transpose(*out_dims).values, sw, rtol=RTOL)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_acc_with_climatology_gather_matches_oracle(backend, layout):
  rng = np.random.default_rng(1)
  dims = LAYOUTS[layout]
  cdims = ('dayofyear', 'hour') + tuple(d for d in dims if d not in ('init_time', 'lead_time'))
  cshape = [366, 4] + [SIZES[d] for d in cdims[2:]]
  clim_vals = (rng.normal(size=cshape) * 10 + 280).astype(np.float32)
  clim = xr.Dataset({'z': xr.DataArray(clim_vals, dims=cdims, coords={
      'dayofyear': np.arange(1, 367), 'hour': np.array([0, 6, 12, 18]), 'latitude': LAT, 'longitude': LON,
      'level': np.array([500, 700, 850])})})
  p, t = _field(rng, dims, np.float32, 280.0), _field(rng, dims, np.float32, 280.0)
  metrics = {'acc': deterministic.ACC(clim), 'rmse': deterministic.RMSE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                               weigh_by=[weighting.GridAreaWeighting()])
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, {'z': p}, {'z': t})
  vt = p['init_time'].values[:, None] + p['lead_time'].values[None, :]
  c, c_dims = O.align_climatology(clim_vals, cdims, vt, ('init_time', 'lead_time'))
  c = O.expand_to(c, c_dims, dims)
  w = (O.grid_area_weights(LAT), ('latitude',))
  means = {}
  for name, arr in (('spa', O.squared_prediction_anomaly(p.values, c)), ('sta', O.squared_target_anomaly(t.values, c)),
                    ('cov', O.anomaly_covariance(p.values, t.values, c)), ('se', O.squared_error(p.values, t.values))):
    sws, sw, out_dims = O.aggregate(arr, dims, ['init_time', 'latitude', 'longitude'], weights=[w])
    means[name] = sws / sw
  np.testing.assert_allclose(res['acc.z'].transpose(*out_dims).values, O.acc(means['cov'], means['spa'], means['sta']),
                             rtol=RTOL)
  np.testing.assert_allclose(res['rmse.z'].transpose(*out_dims).values, O.rmse(means['se']), rtol=RTOL)


@pytest.mark.parametrize('m', [2, 3, 4, 6, 8, 9, 13, 16, 17, 24, 31, 32, 33, 49, 50, 51, 52, 63, 64])
def test_rank_form_spread_for_every_register_bucket(backend, m):
  """The rank form sorts a point's members with a network of compare-exchanges and 3-sorters per register bucket (4 / 8 / 16 /
  32 / 64 padded with +inf, exact 50 / 51: csrc/gen_sortnet3.py).  Heavy ties, reversed and already sorted members, huge and
  tiny magnitudes next to each other: the spread must be the float64 oracle's for every ensemble size around the bucket edges."""
  rng = np.random.default_rng(100 + m)
  nlat, nlon = 8, 48
  lat, lon = np.linspace(-78.75, 78.75, nlat), np.arange(nlon) * 7.5
  pv = np.empty((m, nlat, nlon), np.float32)
  pv[:, 0] = rng.integers(0, 3, size=(m, nlon))                      # many equal members
  pv[:, 1] = np.sort(rng.normal(size=(m, nlon)), axis=0)             # already sorted
  pv[:, 2] = np.sort(rng.normal(size=(m, nlon)), axis=0)[::-1]       # reversed
  pv[:, 3] = rng.normal(size=(m, nlon)) * 10.0 ** rng.integers(-20, 20, size=(m, nlon))  # mixed magnitudes
  pv[:, 4] = np.where(rng.random((m, nlon)) < 0.5, -0.0, 0.0)        # signed zeros
  pv[:, 5] = (rng.integers(-4000, 4000, size=(m, nlon)) * np.float64(1.4e-45)).astype(np.float32)  # fp32 denormals, not flushed
  assert 0 < np.abs(pv[:, 5]).max() < 1.2e-38
  pv[:, 6:] = rng.normal(size=(m, nlat - 6, nlon)) + 280.0
  coords = {'latitude': lat, 'longitude': lon}
  p = {'v': xr.DataArray(pv, dims=('number', 'latitude', 'longitude'), coords=coords)}
  t = {'v': xr.DataArray(pv[0].copy(), dims=('latitude', 'longitude'), coords=coords)}
  agg = aggregation.Aggregator(reduce_dims=['longitude'])
  got = aggregation.compute_metric_values_for_single_chunk({'spread': probabilistic.CRPSSpread(use_sort=True)}, agg, p, t)
  want, _ = O.crps_spread(pv, ('number', 'latitude', 'longitude'), 'number', fair=True, use_sort=True)
  # M = 50 / 51 run the fp32 chain sums (wbx_ens_impl.hpp stats32: worst case 9 * 2^-24 per point, north_star 1e-6); the other
  # buckets and the out-of-range escape (mixed magnitudes, denormals) are fp64 sums
  np.testing.assert_allclose(got['spread.v'].values, want.mean(axis=-1), rtol=1e-6 if m in (50, 51) else 1e-12, atol=1e-300)


@pytest.mark.parametrize('member_layout', ['member_slow', 'member_fast'])
@pytest.mark.parametrize('m,dtype', [(5, np.float32), (51, np.float32), (7, np.float64), (70, np.float32)])
@pytest.mark.parametrize('use_sort', [True, False])
def test_ensemble_suite_matches_oracle(backend, member_layout, m, dtype, use_sort):
  rng = np.random.default_rng(2)
  nlat, nlon = 16, 24
  lat, lon = np.linspace(-84.375, 84.375, nlat), np.arange(nlon) * 15.0
  tv = rng.normal(size=(2, nlat, nlon)).astype(dtype) + 280
  if member_layout == 'member_slow':
    pdims = ('lead_time', 'number', 'latitude', 'longitude')
    pv = (tv[:, None] + rng.normal(size=(2, m, nlat, nlon))).astype(dtype)
  else:
    pdims = ('lead_time', 'latitude', 'longitude', 'number')
    pv = (tv[..., None] + rng.normal(size=(2, nlat, nlon, m))).astype(dtype)
  coords = {'latitude': lat, 'longitude': lon}
  p = {'t2m': xr.DataArray(pv, dims=pdims, coords=coords)}
  t = {'t2m': xr.DataArray(tv, dims=('lead_time', 'latitude', 'longitude'), coords=coords)}
  metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=use_sort),
             'ssr': probabilistic.UnbiasedSpreadSkillRatio(),
             'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')])}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, t)
  tdims = ('lead_time', 'latitude', 'longitude')
  w = (O.grid_area_weights(lat), ('latitude',))

  def mean(arr, d):
    sws, sw, _ = O.aggregate(arr, d, ['latitude', 'longitude'], weights=[w])
    return sws / sw
  skill, sd = O.crps_skill(pv, pdims, tv, tdims, 'number')
  spread, _ = O.crps_spread(pv, pdims, 'number', fair=True, use_sort=use_sort)
  var, _ = O.ensemble_variance(pv, pdims, 'number')
  ue, _ = O.unbiased_ensemble_mean_squared_error(pv, pdims, tv, tdims, 'number')
  em, _ = O.ensemble_mean_squared_error(pv, pdims, tv, tdims, 'number')
  np.testing.assert_allclose(res['crps.t2m'].values, O.crps(mean(skill, sd), mean(spread, sd)), rtol=RTOL)
  np.testing.assert_allclose(res['ssr.t2m'].values, O.unbiased_spread_skill_ratio(mean(var, sd), mean(ue, sd)), rtol=RTOL)
  np.testing.assert_allclose(res['mean_rmse.t2m'].values, np.sqrt(mean(em, sd)), rtol=RTOL)


def test_nan_member_poisons_ensemble_statistics(backend):
  rng = np.random.default_rng(3)
  pv = rng.normal(size=(8, 4, 6)).astype(np.float32)
  pv[3, 1, 2] = np.nan
  p = {'v': xr.DataArray(pv, dims=('number', 'latitude', 'longitude'))}
  t = {'v': xr.DataArray(rng.normal(size=(4, 6)).astype(np.float32), dims=('latitude', 'longitude'))}
  for use_sort in (True, False):
    stats = metrics_base.compute_unique_statistics_for_all_metrics(
        {'crps': probabilistic.CRPSEnsemble(use_sort=use_sort)}, p, t)
    for s in stats.values():
      vals = s['v'].values
      assert np.isnan(vals[1, 2]) and np.isfinite(np.delete(vals.reshape(-1), 1 * 6 + 2)).all()


def test_materialised_statistics_match_oracle(backend):
  rng = np.random.default_rng(4)
  p = xr.DataArray(rng.normal(size=(3, 5, 4)).astype(np.float32), dims=('latitude', 'longitude', 'level'))
  t = xr.DataArray(rng.normal(size=(5, 3)).astype(np.float32), dims=('longitude', 'latitude'))  # broadcast + transposed
  se = deterministic.SquaredError().compute({'v': p}, {'v': t})['v']
  assert se.dims == ('latitude', 'longitude', 'level') and se.shape == (3, 5, 4)
  want = O.squared_error(p.values, O.expand_to(t.values, t.dims, p.dims))
  np.testing.assert_allclose(se.values, want, rtol=1e-12)
  # arithmetic on a lazy statistic materialises it transparently
  np.testing.assert_allclose((se * 2).values, 2 * want, rtol=1e-12)


def test_masked_and_skipna_ensemble_aggregation(backend):
  """masked=True is the public-benchmark default (run_benchmark_evaluation.py:379): the target mask coordinate must
  reach the ensemble statistics that look at the targets (aggregation.py:339-357); the spread term only looks at the
  predictions, carries no mask and stays unmasked, exactly as in the reference."""
  rng = np.random.default_rng(8)
  lat = np.linspace(-80, 80, 9)
  tv = rng.normal(size=(9, 12)).astype(np.float32)
  tv[:3] = np.nan                                   # e.g. sea-ice / SST style missing targets
  pv = (np.nan_to_num(tv)[None] + rng.normal(size=(6, 9, 12))).astype(np.float32)
  t = xr.DataArray(tv, dims=('latitude', 'longitude'), coords={'latitude': lat})
  t.coords['mask'] = ~np.isnan(t)
  p = {'v': xr.DataArray(pv, dims=('number', 'latitude', 'longitude'), coords={'latitude': lat})}
  metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
  w = (O.grid_area_weights(lat), ('latitude',))
  pd, td = ('number', 'latitude', 'longitude'), ('latitude', 'longitude')
  skill = O.crps_skill(pv, pd, tv, td, 'number')[0]
  spread = O.crps_spread(pv, pd, 'number', use_sort=True)[0]
  valid = ~np.isnan(tv)
  for kw, okw in ((dict(masked=True), dict(mask=valid, mask_dims=td)), (dict(skipna=True), dict(skipna=True))):
    agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], **kw)
    res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, {'v': t})
    a = O.aggregate(skill, td, list(td), weights=[w], **okw)
    # the spread is a statistic of the predictions alone: no `mask` coordinate reaches it, so masked=True averages it
    # over every point, like the reference (probabilistic.py:196-247, aggregation.py:339-352)
    b = O.aggregate(spread, td, list(td), weights=[w], **({} if 'mask' in okw else dict(skipna=True)))
    np.testing.assert_allclose(res['crps.v'].values, O.crps(a[0] / a[1], b[0] / b[1]), rtol=RTOL)
  # without either, the NaN targets poison the skill term
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'])
  assert np.isnan(aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, {'v': t})['crps.v'].values)


MANY_REGIONS = {'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360)), 'nh': ((20, 90), (0, 360)),
                'sh': ((-90, -20), (0, 360)), 'europe': ((35, 75), (-12.5, 42.5)), 'namerica': ((25, 60), (240, 285)),
                'ausnz': ((-45, -12.5), (120, 175))}


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest', 'level_fastest'])
@pytest.mark.parametrize('reduce_dims', [('init_time', 'latitude', 'longitude'), ('latitude',), ('init_time',)])
@pytest.mark.parametrize('with_nan', [False, True])
@pytest.mark.parametrize('binned', ['never', 'always'])
def test_many_boolean_bins_use_membership_bits(backend, monkeypatch, layout, reduce_dims, with_nan, binned):
  """>= 5 boolean bins (here 7 regions x {all, land} = 14) go through the bit-mask contraction
  (wbx_contract_bits) or the fused wbx_det_binned kernel; results and the NaN-poisons-every-bin rule must equal
  the dense xr.dot semantics."""
  from weatherbenchx_amd import engine
  monkeypatch.setattr(engine, 'BINNED_MODE', binned)
  rng = np.random.default_rng(11)
  dims = LAYOUTS[layout]
  p, t = _field(rng, dims, np.float32, 280.0), _field(rng, dims, np.float32, 280.0)
  if with_nan:
    idx = {'init_time': 1, 'lead_time': 2, 'level': 0, 'latitude': 30, 'longitude': 5}  # a point in 'nh' only
    p.data[tuple(idx[d] for d in dims)] = np.nan
  land = rng.random((32, 64)) > 0.6
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': LAT, 'longitude': LON})
  agg = aggregation.Aggregator(reduce_dims=list(reduce_dims), weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(MANY_REGIONS, land_sea_mask=lsm)])
  stats = metrics_base.compute_unique_statistics_for_all_metrics({'mse': deterministic.MSE()}, {'z': p}, {'z': t})
  state = agg.aggregate_statistics(stats)
  w = (O.grid_area_weights(LAT), ('latitude',))
  names, masks = O.region_masks(LAT, LON, MANY_REGIONS, land_sea_mask=land)
  sws, sw, out_dims = O.aggregate(O.squared_error(p.values, t.values), dims, reduce_dims, weights=[w],
                                  bin_masks=[('region', masks, ('region', 'latitude', 'longitude'))])
  got = state.sum_weighted_statistics['SquaredError']['z']
  assert list(got['region'].values) == names and len(names) == 14
  np.testing.assert_allclose(got.transpose(*out_dims).values, sws, rtol=RTOL, atol=1e-9)
  np.testing.assert_allclose(state.sum_weights['SquaredError']['z'].transpose(*out_dims).values, sw, rtol=RTOL)
  if with_nan and 'latitude' in reduce_dims and 'longitude' in reduce_dims:
    assert np.isnan(got.values).any()
  # the bit path really was taken
  w_da, _ = agg._cached_weight_product(stats['SquaredError']['z'])
  assert any(v.kind == 'bits' for v in w_da.__dict__['_wbx_w'].values())
  assert engine.BITS_MIN_BINS <= 14 <= 64


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
@pytest.mark.parametrize('mode', ['plain', 'masked', 'skipna'])
def test_fused_binned_kernel_equals_two_stage_path(backend, monkeypatch, layout, mode):
  """The public benchmark's chunk shape (1 init x leads x levels, 34 region x land/sea bins, masked=True;
  run_benchmark_evaluation.py:97-131,369-382) runs through wbx_det_binned: it must agree with the two-stage path
  for every deterministic family (DET3, DET6 with climatology gather, PASS1 wrappers) and the count lanes."""
  from weatherbenchx_amd import engine
  rng = np.random.default_rng(5)
  dims = ('init_time', 'lead_time', 'level') + (('latitude', 'longitude') if layout == 'lon_fastest' else
                                                 ('longitude', 'latitude'))
  sizes = {'init_time': 1, 'lead_time': 3, 'level': 2, 'latitude': 32, 'longitude': 64}
  coords = {'init_time': np.array(['2020-01-01T00'], 'datetime64[ns]'),
            'lead_time': np.arange(3) * np.timedelta64(6, 'h'), 'level': [500, 850], 'latitude': LAT, 'longitude': LON}
  shape = tuple(sizes[d] for d in dims)
  p = xr.DataArray((rng.normal(size=shape) + 280).astype(np.float32), dims=dims, coords=coords)
  tv = (rng.normal(size=shape) + 280).astype(np.float32)
  if mode != 'plain':
    tv[rng.random(shape) < 0.1] = np.nan
  t = xr.DataArray(tv, dims=dims, coords=coords)
  if mode == 'masked':
    t.coords['mask'] = ~np.isnan(t)
  clim = xr.DataArray((rng.normal(size=(366, 4, 2, 32, 64)) + 280).astype(np.float32),
                      dims=('dayofyear', 'hour', 'level', 'latitude', 'longitude'),
                      coords={'dayofyear': np.arange(1, 367), 'hour': [0, 6, 12, 18], 'level': [500, 850],
                              'latitude': LAT, 'longitude': LON})
  land = rng.random((32, 64)) > 0.6
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': LAT, 'longitude': LON})
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'acc': deterministic.ACC({'z': clim}),
             'bias': deterministic.Bias()}
  results, calls = {}, []
  inner = engine._run_binned
  monkeypatch.setattr(engine, '_run_binned', lambda *a, **k: (calls.append(engine.BINNED_MODE), inner(*a, **k))[1])
  for binned in ('never', 'always'):
    monkeypatch.setattr(engine, 'BINNED_MODE', binned)
    engine.clear_caches()
    agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                                 weigh_by=[weighting.GridAreaWeighting()],
                                 bin_by=[binning.Regions(MANY_REGIONS, land_sea_mask=lsm)],
                                 masked=(mode == 'masked'), skipna=(mode == 'skipna'))
    results[binned] = aggregation.compute_metric_values_for_single_chunk(metrics, agg, {'z': p}, {'z': t})
  assert calls and set(calls) == {'always'}
  for k, v in results['never'].items():
    assert v.sizes['region'] == 14
    xr.assert_allclose(results['always'][k], v, rtol=1e-9, atol=1e-12, check_dim_order=False)
  if mode == 'plain':
    w = (O.grid_area_weights(LAT), ('latitude',))
    names, masks = O.region_masks(LAT, LON, MANY_REGIONS, land_sea_mask=land)
    sws, sw, out_dims = O.aggregate(O.squared_error(p.values, t.values), dims, ['init_time', 'latitude', 'longitude'],
                                    weights=[w], bin_masks=[('region', masks, ('region', 'latitude', 'longitude'))])
    np.testing.assert_allclose(results['always']['rmse.z'].transpose(*out_dims).values, np.sqrt(sws / sw), rtol=RTOL)
  else:
    assert np.isfinite(results['always']['rmse.z'].values).all()  # NaN targets are masked / skipped, not propagated


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
@pytest.mark.parametrize('weights', ['area', 'area_x_level', 'none'])
def test_binned_kernel_factored_weights(backend, monkeypatch, layout, weights):
  """wbx_det_binned takes weights that depend on x only (latitude weights on latitude-fastest data) or on the rows only
  (longitude-fastest) in factored form, WBX_BINNED_WT_X_ONLY / WBX_BINNED_WT_ROW_ONLY; weights that also depend on a
  kept dim stay separable per (kept) cell; the dense operand must give the same sums."""
  from weatherbenchx_amd import _hip, engine
  rng = np.random.default_rng(11)
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  dims = ('lead_time', 'level') + sp
  sizes = {'lead_time': 2, 'level': 3, 'latitude': 32, 'longitude': 64}
  coords = {'lead_time': np.arange(2) * np.timedelta64(6, 'h'), 'level': [300, 500, 850], 'latitude': LAT, 'longitude': LON}
  shape = tuple(sizes[d] for d in dims)
  p = xr.DataArray(rng.normal(size=shape).astype(np.float32), dims=dims, coords=coords)
  t = xr.DataArray(rng.normal(size=shape).astype(np.float32), dims=dims, coords=coords)
  lsm = xr.DataArray(rng.random((32, 64)) > 0.5, dims=('latitude', 'longitude'), coords={'latitude': LAT, 'longitude': LON})

  class LevelWeights(weighting.Weighting):
    def weights(self, statistic):
      return xr.DataArray(np.array([1.0, 2.0, 0.5]), dims=('level',), coords={'level': [300, 500, 850]})
  wby = {'area': [weighting.GridAreaWeighting()], 'area_x_level': [weighting.GridAreaWeighting(), LevelWeights()],
         'none': []}[weights]
  monkeypatch.setattr(engine, 'BINNED_MODE', 'always')
  out, flags = {}, {}
  for sep in (True, False):
    monkeypatch.setattr(engine, 'SEPARABLE_BINNED_WEIGHTS', sep)
    engine.clear_caches()
    monkeypatch.setattr(engine, 'S1_EVENT_LOG', [])
    agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=wby,
                                 bin_by=[binning.Regions(MANY_REGIONS, land_sea_mask=lsm)])
    out[sep] = aggregation.compute_metric_values_for_single_chunk({'rmse': deterministic.RMSE()}, agg, {'z': p}, {'z': t})
    flags[sep] = {e.get('w_flags') for e in engine.S1_EVENT_LOG if e['kind'] == 'det_binned'}
    monkeypatch.setattr(engine, 'S1_EVENT_LOG', None)
  np.testing.assert_allclose(out[True]['rmse.z'].values, out[False]['rmse.z'].values, rtol=1e-12)
  if backend != 'emulated':  # the NumPy plan interpreter does not keep an event log
    assert flags[False] == {_hip.BINNED_W_ON_X}
    want = _hip.BINNED_WT_X_ONLY if (layout == 'lat_fastest' or weights == 'none') else _hip.BINNED_WT_ROW_ONLY
    assert flags[True] == {_hip.BINNED_W_ON_X | want}


@pytest.mark.parametrize('m,fair', list(itertools.product([4, 5], [True, False])))
def test_crps_with_nan_member_equals_dropping_it(backend, m, fair):
  # metrics_test.py:1199-1274: skipna_ensemble=True with one all-NaN member == the ensemble without that member
  targets = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True, seed=20)
  predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True,
                                               ensemble_size=m, seed=21)
  with_nan = predictions.copy(deep=True)
  with_nan['2m_temperature'][{'realization': 0}] = np.nan
  kw = dict(ensemble_dim='realization', use_sort=False, fair=fair)
  skip = {'crps': probabilistic.CRPSEnsemble(skipna_ensemble=True, **kw),
          'ssr': probabilistic.UnbiasedSpreadSkillRatio(ensemble_dim='realization', skipna_ensemble=True)}
  plain = {'crps': probabilistic.CRPSEnsemble(skipna_ensemble=False, **kw),
           'ssr': probabilistic.UnbiasedSpreadSkillRatio(ensemble_dim='realization')}
  got = compute_all_metrics(skip, with_nan, targets, ['latitude', 'longitude'])
  dropped = compute_all_metrics(plain, predictions.isel(realization=slice(1, None)), targets, ['latitude', 'longitude'])
  full = compute_all_metrics(plain, predictions, targets, ['latitude', 'longitude'])
  for k in ('crps', 'ssr'):
    xr.assert_allclose(got[f'{k}.2m_temperature'], dropped[f'{k}.2m_temperature'], rtol=1e-9, check_dim_order=False)
    xr.assert_allclose(got[f'{k}.geopotential'], full[f'{k}.geopotential'], rtol=1e-9, check_dim_order=False)
  with pytest.raises(ValueError, match='Failed to compute statistic CRPSSpread') as info:
    compute_all_metrics({'c': probabilistic.CRPSEnsemble(ensemble_dim='realization', use_sort=True,
                                                         skipna_ensemble=True)}, with_nan, targets, ['latitude'])
  assert 'not supported with use_sort=True' in str(info.value.__cause__)


@pytest.mark.parametrize('fair', [True, False])
def test_skipna_ensemble_with_scattered_nan_members_matches_oracle(backend, fair):
  """skipna_ensemble=True where the NaNs differ from point to point (per-point member counts, probabilistic.py:206-216,
  :304-314): skill, pairwise spread, variance and the unbiased mean squared error against the float64 restatement;
  a point with fewer than two members left is NaN and poisons its aggregate exactly as in the reference."""
  rng = np.random.default_rng(44)
  m, nlat, nlon = 6, 19, 36
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * 10.0
  pv = rng.normal(size=(2, m, nlat, nlon)).astype(np.float32)
  tv = rng.normal(size=(2, nlat, nlon)).astype(np.float32)
  pv[rng.random(pv.shape) < 0.15] = np.nan
  pv[1, :, 3, 4] = np.nan   # a point without any member
  pv[1, 1:, 5, 6] = np.nan  # a point with a single member
  pdims, tdims = ('time', 'number', 'latitude', 'longitude'), ('time', 'latitude', 'longitude')
  coords = {'latitude': lat, 'longitude': lon}
  p = {'v': xr.DataArray(pv, dims=pdims, coords=coords)}
  t = {'v': xr.DataArray(tv, dims=tdims, coords=coords)}
  stats = {'skill': probabilistic.CRPSSkill(skipna_ensemble=True),
           'spread': probabilistic.CRPSSpread(use_sort=False, fair=fair, skipna_ensemble=True),
           'var': probabilistic.EnsembleVariance(skipna_ensemble=True),
           'uemse': probabilistic.UnbiasedEnsembleMeanSquaredError(skipna_ensemble=True)}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], skipna=True)
  got = aggregation.compute_metric_values_for_single_chunk(stats, agg, p, t)
  w = (O.grid_area_weights(lat), ('latitude',))
  with np.errstate(invalid='ignore', divide='ignore'):
    want = {'skill': O.crps_skill(pv, pdims, tv, tdims, 'number', skipna_ensemble=True),
            'spread': O.crps_spread(pv, pdims, 'number', fair=fair, skipna_ensemble=True),
            'var': O.ensemble_variance(pv, pdims, 'number', skipna_ensemble=True),
            'uemse': O.unbiased_ensemble_mean_squared_error(pv, pdims, tv, tdims, 'number', skipna_ensemble=True)}
    for k, (vals, dims) in want.items():
      sws, sw, _ = O.aggregate(vals, dims, ['latitude', 'longitude'], weights=[w], skipna=True)
      np.testing.assert_allclose(got[f'{k}.v'].values, sws / sw, rtol=RTOL, err_msg=k)


@pytest.mark.parametrize('m,use_sort,fair', list(itertools.product([4, 5], [False, True], [True, False])))
def test_crps_ensemble_distance(backend, m, use_sort, fair):
  # metrics_test.py:673-752
  targets = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True,
                                           ensemble_size=m + 1, seed=0)
  predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', random=True,
                                               ensemble_size=m, seed=1)
  no_ens = targets.isel(realization=0, drop=True)
  no_spread = no_ens.expand_dims(realization=np.arange(m + 1))  # ensemble dim first: a different layout on purpose
  dist = {'crps': probabilistic.CRPSEnsembleDistance(ensemble_dim='realization', use_sort=use_sort, fair=fair)}
  plain = {'crps': probabilistic.CRPSEnsemble(ensemble_dim='realization', use_sort=use_sort, fair=fair)}
  rd = ['latitude', 'longitude', 'time']
  a = compute_all_metrics(dist, predictions, targets, rd)
  b = compute_all_metrics(dist, predictions, no_spread, rd)
  c = compute_all_metrics(plain, predictions, no_ens, rd)
  stderr = 1 / np.sqrt(np.prod([m * targets['geopotential'].sizes[d] for d in rd]))
  for v in ['2m_temperature', 'geopotential']:
    if fair:
      np.testing.assert_allclose(a[f'crps.{v}'].values, 0, atol=5 * stderr)
    xr.assert_allclose(b[f'crps.{v}'], c[f'crps.{v}'], atol=5 * stderr, check_dim_order=False)
  # brute force for one variable (float64 oracle, explicit double loop)
  p = predictions['2m_temperature'].values.astype(np.float64)
  t = targets['2m_temperature'].values.astype(np.float64)
  skill = np.abs(p[..., :, None] - t[..., None, :]).mean(axis=(-1, -2))
  sp = np.abs(p[..., :, None] - p[..., None, :]).sum(axis=(-1, -2)) / (m * (m - int(fair)))
  st = np.abs(t[..., :, None] - t[..., None, :]).sum(axis=(-1, -2)) / ((m + 1) * (m + 1 - int(fair)))
  want = (skill - 0.5 * sp - 0.5 * st).mean(axis=(1, 2, 3))  # dims: prediction_timedelta, time, lat, lon
  np.testing.assert_allclose(a['crps.2m_temperature'].values, want, rtol=1e-9)


@pytest.mark.parametrize('np_members,nt_members', [(5, 3), (4, 2)])
def test_unbiased_mse_with_an_ensemble_of_targets(backend, np_members, nt_members):
  """probabilistic.py:320-336: (mean p - mean t)^2 - var_p / M - var_t / N per point, area-weighted.  Here it is
  the mean over target members of the fused UEMSE lanes minus the target-variance lane (LinearCombination)."""
  rng = np.random.default_rng(31)
  lat = np.linspace(-80, 80, 9)
  pv = rng.normal(size=(np_members, 2, 9, 12)).astype(np.float32) + 2.0
  tv = rng.normal(size=(2, nt_members, 9, 12)).astype(np.float32)
  p = {'v': xr.DataArray(pv, dims=('number', 'lead_time', 'latitude', 'longitude'), coords={'latitude': lat})}
  t = {'v': xr.DataArray(tv, dims=('lead_time', 'number', 'latitude', 'longitude'), coords={'latitude': lat})}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  got = aggregation.compute_metric_values_for_single_chunk({'u': probabilistic.UnbiasedEnsembleMeanRMSE()}, agg, p, t)
  p64, t64 = pv.astype(np.float64), tv.astype(np.float64)
  pdims, tdims = ('number', 'lead_time', 'latitude', 'longitude'), ('lead_time', 'number', 'latitude', 'longitude')
  stat, sdims = O.unbiased_ensemble_mean_squared_error(pv, pdims, tv, tdims, 'number')  # (pinned: test_oracle_reference_pins.py)
  assert sdims == ('lead_time', 'latitude', 'longitude')
  w = O.grid_area_weights(lat)[None, :, None]
  want = np.sqrt((stat * w).sum(axis=(1, 2)) / (np.broadcast_to(w, stat.shape).sum(axis=(1, 2))))
  np.testing.assert_allclose(got['u.v'].values, want, rtol=RTOL)
  # the un-fused route (materialised statistic) agrees
  stats = metrics_base.compute_unique_statistics_for_all_metrics({'u': probabilistic.UnbiasedEnsembleMeanRMSE()}, p, t)
  (name, per_var), = stats.items()
  np.testing.assert_allclose(per_var['v'].values, stat, rtol=1e-5, atol=1e-6)
  # skipna_ensemble=True with NaN members on BOTH sides (probabilistic.py:304-333, 133-145): per-point member counts,
  # evaluated un-fused and reduced through the generic kernels
  pn, tn = pv.copy(), tv.copy()
  pn[0, :, 2:4] = np.nan   # a prediction member missing over a band
  tn[:, 1, 5:7] = np.nan   # a target member missing over another band
  pn[1, 0, 0, 0] = np.nan
  pq = {'v': xr.DataArray(pn, dims=('number', 'lead_time', 'latitude', 'longitude'), coords={'latitude': lat})}
  tq = {'v': xr.DataArray(tn, dims=('lead_time', 'number', 'latitude', 'longitude'), coords={'latitude': lat})}
  got = aggregation.compute_metric_values_for_single_chunk(
      {'u': probabilistic.UnbiasedEnsembleMeanRMSE(skipna_ensemble=True)}, agg, pq, tq)
  p64, t64 = pn.astype(np.float64), tn.astype(np.float64)
  stat, _ = O.unbiased_ensemble_mean_squared_error(pn, pdims, tn, tdims, 'number', skipna_ensemble=True)
  want = np.sqrt((stat * w).sum(axis=(1, 2)) / (np.broadcast_to(w, stat.shape).sum(axis=(1, 2))))
  np.testing.assert_allclose(got['u.v'].values, want, rtol=RTOL)
  # CRPSSkill: mean over the non-NaN (prediction member, target member) pairs of each point
  skill = probabilistic.CRPSSkill(skipna_ensemble=True).compute(pq, tq)['v']
  tt = np.moveaxis(t64, 1, 0)                                     # [N, lead, lat, lon]
  pairs = np.abs(p64[:, None] - tt[None])                          # [M, N, lead, lat, lon]
  with np.errstate(invalid='ignore'):
    want_skill = np.nanmean(pairs.reshape(-1, *pairs.shape[2:]), axis=0)
  np.testing.assert_allclose(skill.transpose('lead_time', 'latitude', 'longitude').values, want_skill, rtol=1e-6)
  assert np.isclose(O.crps_skill(pn, ('number', 'lead_time', 'latitude', 'longitude'), tn,
                                 ('lead_time', 'number', 'latitude', 'longitude'), 'number', skipna_ensemble=True)[0],
                    want_skill, rtol=1e-12).all()  # the oracle agrees with the brute force


@pytest.mark.parametrize('nlon,expect_rows', [(24, 8), (25, 5), (29, 0)])
def test_latitude_fastest_plane_mode_matches_oracle(backend, nlon, expect_rows):
  """Latitude-fastest chunks with >= 64 latitudes take the LDS 'plane mode' kernel (aligned 16-B loads of R-row
  spans) when R divides the number of longitudes; results must equal the oracle exactly like the generic path."""
  from weatherbenchx_amd import engine, planner
  rng = np.random.default_rng(12)
  nlat = 91
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * (360.0 / nlon)
  dims = ('init_time', 'lead_time', 'level', 'longitude', 'latitude')
  shape = (3, 2, 2, nlon, nlat)
  coords = {'latitude': lat, 'longitude': lon, 'level': np.array([500, 850]),
            'init_time': np.array(['2020-01-01T00', '2020-01-02T00', '2020-01-03T00'], dtype='datetime64[ns]'),
            'lead_time': (np.arange(2) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')}
  p = xr.DataArray((rng.normal(size=shape) + 280).astype(np.float32), dims=dims, coords=coords)
  t = xr.DataArray((rng.normal(size=shape) + 280).astype(np.float32), dims=dims, coords=coords)
  cdims = ('dayofyear', 'hour', 'level', 'longitude', 'latitude')
  cv = (rng.normal(size=(5, 4, 2, nlon, nlat)) * 10 + 280).astype(np.float32)
  clim = xr.Dataset({'z': xr.DataArray(cv, dims=cdims, coords={'dayofyear': np.arange(1, 6), 'hour': np.array([0, 6, 12, 18]),
                                                               'level': coords['level'], 'longitude': lon, 'latitude': lat})})
  metrics = {'acc': deterministic.ACC(clim), 'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, {'z': p}, {'z': t})
  vt = coords['init_time'][:, None] + coords['lead_time'][None, :]
  c, cd = O.align_climatology(cv, cdims, vt, ('init_time', 'lead_time'))
  c = O.expand_to(c, cd, dims)
  w = (O.grid_area_weights(lat), ('latitude',))

  def mean(a):
    sws, sw, od = O.aggregate(a, dims, ['init_time', 'latitude', 'longitude'], weights=[w])
    return sws / sw
  np.testing.assert_allclose(res['rmse.z'].values, np.sqrt(mean(O.squared_error(p.values, t.values))), rtol=RTOL)
  np.testing.assert_allclose(res['bias.z'].values, mean(O.error(p.values, t.values)), rtol=RTOL, atol=1e-9)
  want = O.acc(mean(O.anomaly_covariance(p.values, t.values, c)), mean(O.squared_prediction_anomaly(p.values, c)),
               mean(O.squared_target_anomaly(t.values, c)))
  np.testing.assert_allclose(res['acc.z'].values, want, rtol=RTOL)
  # the planner really chose (or rejected) plane mode
  lays = [planner.InputLayout(strides=dict(zip(dims, [int(s // 4) for s in a.values.strides])), itemsize=4,
                              base_alignment=256) for a in (p, t)]
  plan = planner.build_s1_plan(dims, dict(zip(dims, shape)), lays, ['init_time', 'latitude', 'longitude'],
                               wdep_dims=['latitude'])
  assert plan.x_dim == 'latitude' and plan.x_kept and plan.plane_rows == expect_rows
  assert plan.depth_chunk % max(plan.plane_rows, 1) == 0


def test_user_defined_climatology_statistic_gets_an_aligned_array(backend):
  """Plugins subclassing PerVariableStatisticWithClimatology receive something that behaves like the aligned
  climatology DataArray of the reference (metrics/base.py:397-406)."""
  class AnomalyError(metrics_base.PerVariableStatisticWithClimatology):
    def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
      assert set(aligned_climatology.dims) <= set(predictions.dims)
      return abs((predictions - aligned_climatology) - (targets - aligned_climatology))

  rng = np.random.default_rng(13)
  dims = LAYOUTS['lon_fastest']
  p, t = _field(rng, dims, np.float32), _field(rng, dims, np.float32)
  cdims = ('dayofyear', 'hour', 'level', 'latitude', 'longitude')
  clim = xr.Dataset({'z': xr.DataArray(rng.normal(size=(366, 4, 3, 32, 64)).astype(np.float32), dims=cdims, coords={
      'dayofyear': np.arange(1, 367), 'hour': np.array([0, 6, 12, 18]), 'level': np.array([500, 700, 850]),
      'latitude': LAT, 'longitude': LON})})
  res = compute_all_metrics({'ae': AnomalyError(clim), 'mae': deterministic.MAE()}, {'z': p}, {'z': t},
                            ['latitude', 'longitude'])
  xr.assert_allclose(res['ae.z'], res['mae.z'], rtol=1e-5, check_dim_order=False)  # fp32 plugin arithmetic vs fused fp64


def test_prediction_and_target_passthrough(backend):
  # metrics_test.py:1008-1029
  predictions = xr.DataArray(np.array([[1.0, 2.0], [np.nan, 4.0]]), dims=['x', 'y'])
  targets = xr.DataArray(np.array([[5.0, np.nan], [7.0, 8.0]]), dims=['x', 'y'])
  r = deterministic.PredictionPassthrough(copy_nans_from_targets=False)._compute_per_variable(predictions, targets)
  np.testing.assert_array_equal(r.values, [[1.0, 2.0], [np.nan, 4.0]])
  r = deterministic.PredictionPassthrough(copy_nans_from_targets=True)._compute_per_variable(predictions, targets)
  np.testing.assert_array_equal(r.values, [[1.0, np.nan], [np.nan, 4.0]])
  r = deterministic.TargetPassthrough(copy_nans_from_predictions=True)._compute_per_variable(predictions, targets)
  np.testing.assert_array_equal(r.values, [[5.0, np.nan], [np.nan, 8.0]])
  # as metrics (PredictionAverage / TargetAverage) they go through the PASS1 family
  res = compute_all_metrics({'pa': deterministic.PredictionAverage(), 'ta': deterministic.TargetAverage()},
                            {'v': predictions.fillna(0.0)}, {'v': targets.fillna(0.0)}, ['x', 'y'])
  np.testing.assert_allclose([res['pa.v'].values, res['ta.v'].values], [7.0 / 4, 20.0 / 4])


def test_error_exceedance_reference_table(backend):
  # metrics_test.py:1031-1046 through the product: per-point values (nothing reduced)
  predictions = xr.DataArray(np.array([0, -1, 1, np.nan]), dims=['x'], name='v')
  targets = xr.DataArray(np.array([0, 0, 0, 0.0]), dims=['x'], name='v')
  result = deterministic.ErrorExceedance(thresholds=xr.DataArray([0, 0.5, 1, np.nan], dims=['y']))._compute_per_variable(
      predictions, targets)
  assert result.dims == ('x', 'y')
  np.testing.assert_array_equal(result.values, np.array([[0, 0, 0, np.nan], [1, 1, 0, np.nan], [1, 1, 0, np.nan],
                                                         [np.nan] * 4]))


def test_rank_histogram_reference_table(backend):
  # metrics_test.py:1310-1370: per element (reduce_dims=[]) and aggregated over (batch, space)
  p = {'geopotential': xr.DataArray(
      np.array([[[0.6, 0.2], [0.7, 0.3], [0.8, 0.4], [0.9, 0.5], [1.0, 0.6]],
                [[0.7, 0.6], [0.8, 0.7], [0.9, 0.8], [1.0, 0.9], [1.1, 1.0]]]), dims=['batch', 'number', 'space'])}
  t = {'geopotential': xr.DataArray(np.array([[0.55, 0.65], [0.75, 0.85]]), dims=['batch', 'space'])}
  metrics = {'rank_histogram': probabilistic.RankHistogram()}
  want = np.array([[[1., 0., 0., 0., 0., 0.], [0., 0., 0., 0., 0., 1.]],
                   [[0., 1., 0., 0., 0., 0.], [0., 0., 0., 1., 0., 0.]]])
  per_element = compute_all_metrics(metrics, p, t, [])['rank_histogram.geopotential']
  assert list(per_element['rank'].values) == [0, 1, 2, 3, 4, 5]
  np.testing.assert_allclose(per_element.transpose('batch', 'space', 'rank').values, want)
  aggregated = compute_all_metrics(metrics, p, t, ['batch', 'space'])['rank_histogram.geopotential']
  np.testing.assert_allclose(aggregated.values, want.mean(axis=(0, 1)))


@pytest.mark.parametrize('mode', ['plain', 'masked', 'skipna'])
@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_indicator_statistics_match_oracle(backend, mode, layout):
  """ErrorExceedance / EnsembleErrorExceedance / RankHistogram, area-weighted and binned by region, against the
  oracle; NaN thresholds, NaN members (skipped in the member mean) and NaN targets (masked / skipna) included."""
  rng = np.random.default_rng(17)
  m = 5
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  tdims = ('lead_time',) + sp
  pdims = ('lead_time', 'number') + sp
  n = {'lead_time': 3, 'number': m, 'latitude': 32, 'longitude': 64}
  tv = rng.normal(size=[n[d] for d in tdims]).astype(np.float32)
  pv = (rng.normal(size=[n[d] for d in pdims]) * 1.5).astype(np.float32)
  pv[0, 1] = np.nan                       # one member is all-NaN at lead 0: skipped by the member mean
  if mode != 'plain':
    tv[rng.random(tv.shape) < 0.1] = np.nan
  coords = {'latitude': LAT, 'longitude': LON}
  t = xr.DataArray(tv, dims=tdims, coords=coords)
  if mode == 'masked':
    t.coords['mask'] = ~np.isnan(t)
  p = xr.DataArray(pv, dims=pdims, coords=coords)
  thresholds = [0.5, np.nan, 2.0]
  regions = {'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360))}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(regions)], masked=(mode == 'masked'), skipna=(mode == 'skipna'))
  metrics = {'exc': probabilistic.EnsembleErrorExceedance(thresholds), 'rank': probabilistic.RankHistogram(),
             'det_exc': deterministic.ErrorExceedance(thresholds)}
  p0 = xr.DataArray(pv[:, 0], dims=tdims, coords=coords)
  got = aggregation.compute_metric_values_for_single_chunk({k: metrics[k] for k in ('exc', 'rank')}, agg, {'v': p},
                                                           {'v': t})
  got_det = aggregation.compute_metric_values_for_single_chunk({'det_exc': metrics['det_exc']}, agg, {'v': p0}, {'v': t})
  w = (O.grid_area_weights(LAT), ('latitude',))
  names, masks = O.region_masks(LAT, LON, regions)
  bm = [('region', masks, ('region', 'latitude', 'longitude'))]
  kw = {}
  if mode == 'masked':
    kw = dict(mask=~np.isnan(tv), mask_dims=tdims)
  if mode == 'skipna':
    kw = dict(skipna=True)

  def mean_of(stat, dims):
    sws, sw, od = O.aggregate(stat, dims, ['latitude', 'longitude'], weights=[w], bin_masks=bm, **kw)
    with np.errstate(all='ignore'):
      return sws / sw, od

  want, od = mean_of(*O.ensemble_error_exceedance(pv, pdims, tv, tdims, thresholds, 'number'))
  np.testing.assert_allclose(got['exc.v'].transpose(*od).values, want, rtol=RTOL, equal_nan=True)
  want, od = mean_of(*O.rank_histogram(pv, pdims, tv, tdims, 'number'))
  np.testing.assert_allclose(got['rank.v'].transpose(*od).values, want, rtol=RTOL, equal_nan=True)
  np.testing.assert_allclose(np.nansum(want, axis=od.index('rank')), 1.0)  # a histogram
  want, od = mean_of(*O.error_exceedance(pv[:, 0], tdims, tv, tdims, thresholds))
  np.testing.assert_allclose(got_det['det_exc.v'].transpose(*od).values, want, rtol=RTOL, equal_nan=True)


@pytest.mark.parametrize('mask_order', ['lon_lat', 'lat_lon'])
def test_plane_mode_with_a_validity_mask(backend, mask_order):
  """Latitude-fastest chunk + masked=True with a mask coordinate on the targets (SST-style NaNs): when the mask is
  stored in the data's (longitude, latitude) order its spans ride through LDS with the data (plane mode); a
  (latitude, longitude) mask takes the generic x-kept kernel.  Both must equal the oracle."""
  from weatherbenchx_amd import planner
  rng = np.random.default_rng(21)
  nlat, nlon = 91, 24
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * 15.0
  dims = ('init_time', 'lead_time', 'longitude', 'latitude')
  shape = (3, 2, nlon, nlat)
  coords = {'latitude': lat, 'longitude': lon,
            'init_time': np.array(['2020-01-01T00', '2020-01-02T00', '2020-01-03T00'], dtype='datetime64[ns]'),
            'lead_time': (np.arange(2) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')}
  pv = (rng.normal(size=shape) + 280).astype(np.float32)
  tv = (rng.normal(size=shape) + 280).astype(np.float32)
  valid = rng.random((nlon, nlat)) > 0.3
  tv[:, :, ~valid] = np.nan
  pv[0, 0, ~valid] = np.inf  # masked-out points contribute exactly 0 whatever they hold
  p = xr.DataArray(pv, dims=dims, coords=coords)
  t = xr.DataArray(tv, dims=dims, coords=coords)
  if mask_order == 'lon_lat':
    t.coords['mask'] = xr.DataArray(valid, dims=('longitude', 'latitude'))
  else:
    t.coords['mask'] = xr.DataArray(np.ascontiguousarray(valid.T), dims=('latitude', 'longitude'))
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                               weigh_by=[weighting.GridAreaWeighting()], masked=True)
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, {'z': p}, {'z': t})
  w = (O.grid_area_weights(lat), ('latitude',))
  full_valid = np.broadcast_to(valid, shape)
  with np.errstate(invalid='ignore'):
    se, e = O.squared_error(pv, tv), O.error(pv, tv)
  sws, sw, _ = O.aggregate(se, dims, ['init_time', 'latitude', 'longitude'], weights=[w], mask=full_valid, mask_dims=dims)
  np.testing.assert_allclose(res['rmse.z'].values, np.sqrt(sws / sw), rtol=RTOL)
  sws, sw, _ = O.aggregate(e, dims, ['init_time', 'latitude', 'longitude'], weights=[w], mask=full_valid, mask_dims=dims)
  np.testing.assert_allclose(res['bias.z'].values, sws / sw, rtol=RTOL, atol=1e-9)
  # latitude kept (zonal statistics): nothing can be folded, the LDS plane kernel carries the mask spans itself
  agg_k = aggregation.Aggregator(reduce_dims=['init_time', 'longitude'], masked=True)
  res_k = aggregation.compute_metric_values_for_single_chunk({'mse': deterministic.MSE()}, agg_k, {'z': p}, {'z': t})
  sws, sw, od = O.aggregate(se, dims, ['init_time', 'longitude'], mask=full_valid, mask_dims=dims)
  with np.errstate(invalid='ignore', divide='ignore'):
    np.testing.assert_allclose(res_k['mse.z'].transpose(*od).values, sws / sw, rtol=RTOL, equal_nan=True)
  lays = [planner.InputLayout(strides=dict(zip(dims, [int(s // 4) for s in a.strides])), itemsize=4, base_alignment=256)
          for a in (pv, tv)] + [None]
  mdims = ('longitude', 'latitude') if mask_order == 'lon_lat' else ('latitude', 'longitude')
  mshape = (nlon, nlat) if mask_order == 'lon_lat' else (nlat, nlon)
  lays.append(planner.InputLayout(strides=dict(zip(mdims, (mshape[1], 1))), itemsize=1, base_alignment=256))
  plan = planner.build_s1_plan(dims, dict(zip(dims, shape)), lays, ['init_time', 'latitude', 'longitude'],
                               wdep_dims=['latitude'], flags=1)
  assert plan.x_dim == 'latitude' and plan.x_kept
  assert plan.plane_rows == (8 if mask_order == 'lon_lat' else 0)


@pytest.mark.parametrize('mode', ['plain', 'masked', 'skipna'])
def test_latitude_weights_folded_into_stage_one(backend, monkeypatch, mode):
  """Latitude-fastest data + GridAreaWeighting and no bins: the weights depend on the innermost dim only, so the
  deterministic family applies them inside stage 1 (plan.x_weights, flat float4 sweep over the contiguous planes) and
  sums latitude there, also under a validity mask stored like the data; the ensemble family does the same one point per
  lane (s1_xf1_kernel), masks and skipna included.  The deterministic family under skipna keeps latitude for stage 2
  (measured faster).  Every route must equal the un-folded one and the oracle."""
  from weatherbenchx_amd import engine
  rng = np.random.default_rng(33)
  nlat, nlon, m = 91, 24, 5
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * 15.0
  dims = ('lead_time', 'longitude', 'latitude')
  edims = ('lead_time', 'number', 'longitude', 'latitude')
  coords = {'latitude': lat, 'longitude': lon, 'lead_time': (np.arange(3) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')}
  tv = (rng.normal(size=(3, nlon, nlat)) + 280).astype(np.float32)
  pv = (rng.normal(size=(3, nlon, nlat)) + 280).astype(np.float32)
  ev = (tv[:, None] + rng.normal(size=(3, m, nlon, nlat))).astype(np.float32)
  if mode != 'plain':
    tv[rng.random(tv.shape) < 0.1] = np.nan
  t = xr.DataArray(tv, dims=dims, coords=coords)
  if mode == 'masked':
    t.coords['mask'] = ~np.isnan(t)
  p = xr.DataArray(pv, dims=dims, coords=coords)
  e = xr.DataArray(ev, dims=edims, coords=coords)
  det = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  ens = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
  logs, results = {}, {}
  for fold in (True, False):
    monkeypatch.setattr(engine, 'FOLD_X_WEIGHTS', fold)
    engine.clear_caches()
    seen = []
    inner = engine._planned
    monkeypatch.setattr(engine, '_planned', lambda *a, **k: (seen.append(inner(*a, **k)[0]), inner(*a, **k))[1])
    agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                                 masked=(mode == 'masked'), skipna=(mode == 'skipna'))
    results[fold] = dict(aggregation.compute_metric_values_for_single_chunk(det, agg, {'v': p}, {'v': t}))
    results[fold].update(aggregation.compute_metric_values_for_single_chunk(ens, agg, {'v': e}, {'v': t}))
    logs[fold] = seen
    monkeypatch.setattr(engine, '_planned', inner)
  used = [pl for pl in logs[True] if pl.x_weights is not None]
  assert used and all(not pl.x_kept and pl.plane_rows == nlon for pl in used)
  # deterministic + ensemble plans; skipna keeps the deterministic family on the x-kept kernel (measured faster)
  # (masked: the ensemble lanes that look at the targets are masked, the member-only ones are a second, unmasked launch)
  assert len({id(pl) for pl in used}) == {'plain': 2, 'masked': 3, 'skipna': 1}[mode]
  assert logs[False] and all(pl.x_weights is None and pl.x_kept for pl in logs[False])
  for k, v in results[False].items():
    np.testing.assert_allclose(results[True][k].values, v.values, rtol=1e-9, equal_nan=True)
  w = (O.grid_area_weights(lat), ('latitude',))
  kw = {}
  if mode == 'masked':
    kw = dict(mask=~np.isnan(tv), mask_dims=dims)
  if mode == 'skipna':
    kw = dict(skipna=True)
  with np.errstate(invalid='ignore'):
    sws, sw, _ = O.aggregate(O.squared_error(pv, tv), dims, ['latitude', 'longitude'], weights=[w], **kw)
  np.testing.assert_allclose(results[True]['rmse.v'].values, np.sqrt(sws / sw), rtol=RTOL, equal_nan=True)
  if mode == 'plain':
    sk = O.aggregate(O.crps_skill(ev, edims, tv, dims, 'number')[0], dims, ['latitude', 'longitude'], weights=[w])
    sp = O.aggregate(O.crps_spread(ev, edims, 'number', use_sort=True)[0], dims, ['latitude', 'longitude'], weights=[w])
    np.testing.assert_allclose(results[True]['crps.v'].values, O.crps(sk[0] / sk[1], sp[0] / sp[1]), rtol=RTOL)


def test_error_exceedance_with_thresholds_that_vary_with_data_dims(backend):
  """deterministic.py:262-295 with a thresholds DataArray that carries a data dim (one threshold set per level): the
  comparison broadcasts like `abs_error > thresholds`; NaN errors and NaN thresholds stay NaN."""
  rng = np.random.default_rng(17)
  lat = np.linspace(-80, 80, 9)
  pv = rng.normal(size=(3, 2, 9, 12)).astype(np.float32)
  tv = rng.normal(size=(3, 2, 9, 12)).astype(np.float32)
  tv[0, 0, 0, 0] = np.nan
  dims = ('lead_time', 'level', 'latitude', 'longitude')
  coords = {'level': np.array([500, 850]), 'latitude': lat}
  thr = xr.DataArray(np.array([[0.5, 1.0, np.nan], [1.0, 2.0, 3.0]]), dims=('level', 'error_exceedance_thresholds'),
                     coords={'level': np.array([500, 850]), 'error_exceedance_thresholds': np.array([0, 1, 2])})
  stat = deterministic.ErrorExceedance(thr).compute({'v': xr.DataArray(pv, dims=dims, coords=coords)},
                                                   {'v': xr.DataArray(tv, dims=dims, coords=coords)})['v']
  ae = np.abs(pv.astype(np.float64) - tv.astype(np.float64))[..., None]
  th = thr.values[None, :, None, None, :]
  with np.errstate(invalid='ignore'):
    want = np.where(np.isnan(ae) | np.isnan(th), np.nan, (ae > th).astype(np.float64))
  np.testing.assert_array_equal(stat.transpose(*dims, 'error_exceedance_thresholds').values, want)
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], skipna=True)
  state = agg.aggregate_stat_var(stat)
  got = state.mean_statistics().transpose('lead_time', 'level', 'error_exceedance_thresholds').values
  w = O.grid_area_weights(lat)[None, None, :, None, None]
  ok = ~np.isnan(want)
  with np.errstate(invalid='ignore'):
    ref = (np.where(ok, want, 0) * w).sum(axis=(2, 3)) / (ok * w).sum(axis=(2, 3))
  np.testing.assert_allclose(got, ref, rtol=RTOL, equal_nan=True)
