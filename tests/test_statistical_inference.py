"""t-tests, baseline comparisons and the delta-method linearisation (weatherbenchx_amd/statistical_inference/): the properties the
reference's tests check (statistical_inference/t_test_test.py:24-254: coverage of the intervals for i.i.d. and AR(2) data,
p-values consistent with intervals, constant series) with smaller replicate counts, plus exact pins the reference does not have:
the i.i.d. test against scipy.stats.ttest_1samp / ttest_rel replicate by replicate, the cosine estimator against its definition
written out, the finite-difference linearisation against analytic gradients, and an inference on accumulators that came off
the device."""
import numpy as np
import pytest
import scipy.stats

from weatherbenchx_amd import aggregation
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.statistical_inference import autodiff
from weatherbenchx_amd.statistical_inference import baseline_comparison
from weatherbenchx_amd.statistical_inference import t_test
from weatherbenchx_amd.statistical_inference import utils


class MeanPrediction(metrics_base.Statistic):

  def compute(self, predictions, targets):
    return predictions


class MeanTarget(metrics_base.Statistic):

  def compute(self, predictions, targets):
    return targets


class RatioOfMeans(metrics_base.PerVariableMetric):

  @property
  def statistics(self):
    return {'mean_prediction': MeanPrediction(), 'mean_target': MeanTarget()}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return statistic_values['mean_prediction'] / statistic_values['mean_target']


def _state(per_statistic, weights=None):
  """An AggregationState that no reduction has touched yet: the values themselves, unit weights (or the given ones)."""
  sws, sw = {}, {}
  for stat, da in per_statistic.items():
    w = xr.DataArray(np.ones(da.shape), dims=da.dims, coords=dict(da.coords)) if weights is None else weights
    sws[stat] = {'variable': da * w}
    sw[stat] = {'variable': w}
  return aggregation.AggregationState(sws, sw)


def _mean_setup(data, dims):
  da = xr.DataArray(data, dims=dims, coords={dims[0]: np.arange(data.shape[0])})
  return {'mean': MeanPrediction()}, _state({'MeanPrediction': da})


def simulate_ar2(rng, mean, sigma, phi1, phi2, steps, replicates):
  """A stationary Gaussian AR(2) process started from its stationary distribution."""
  denom = (1 + phi2) * (1 - phi1 ** 2 + phi2 ** 2 - 2 * phi2)
  gamma0, gamma1 = sigma ** 2 * (1 - phi2) / denom, sigma ** 2 * phi1 / denom
  rho1 = gamma1 / gamma0
  x0, x1 = rng.normal(size=replicates), rng.normal(size=replicates)
  out = [np.sqrt(gamma0) * x0, np.sqrt(gamma0) * (rho1 * x0 + np.sqrt(1 - rho1 ** 2) * x1)]
  for _ in range(steps - 2):
    out.append(phi1 * out[-1] + phi2 * out[-2] + rng.normal(size=replicates) * sigma)
  return np.stack(out) + mean


def _coverage(inference, true_value, alpha, name='mean'):
  lower, upper = inference.confidence_intervals(alpha)
  lower, upper = np.asarray(lower[name]['variable'].values), np.asarray(upper[name]['variable'].values)
  return float(((lower <= true_value) & (true_value <= upper)).mean())


def _assert_p_values_consistent_with_intervals(inference, null_value, name='mean'):
  p = np.asarray(inference.p_values(null_value)[name]['variable'].values)
  for alpha in (0.2, 0.05):
    lower, upper = inference.confidence_intervals(alpha)
    inside = (np.asarray(lower[name]['variable'].values) <= null_value) & (null_value <= np.asarray(upper[name]['variable'].values))
    np.testing.assert_array_equal(p > alpha, inside)
    np.testing.assert_array_equal(np.asarray(inference.significance_tests(null_value, alpha)[name]['variable'].values), p <= alpha)


def test_plain_t_test_is_scipys():
  rng = np.random.default_rng(0)
  data = rng.normal(size=(10, 20000)) + 10.0
  metrics, state = _mean_setup(data, ('samples', 'replicates'))
  inference = t_test.IID(metrics=metrics, aggregated_statistics=state, experimental_unit_dim='samples')
  ref = scipy.stats.ttest_1samp(data, popmean=9.5, axis=0)
  np.testing.assert_allclose(np.asarray(inference.p_values(9.5)['mean']['variable'].values), ref.pvalue, rtol=1e-9, atol=1e-15)
  np.testing.assert_allclose(np.asarray(inference.point_estimates()['mean']['variable'].values), data.mean(axis=0), rtol=1e-13)
  np.testing.assert_allclose(np.asarray(inference.standard_error_estimates()['mean']['variable'].values),
                             data.std(axis=0, ddof=1) / np.sqrt(10), rtol=1e-9)
  ci = ref.confidence_interval(0.9)
  lower, upper = inference.confidence_intervals(0.1)
  np.testing.assert_allclose(np.asarray(lower['mean']['variable'].values), ci.low, rtol=1e-9)
  np.testing.assert_allclose(np.asarray(upper['mean']['variable'].values), ci.high, rtol=1e-9)
  for alpha in (0.2, 0.1, 0.05):                                        # t_test_test.py:24-54: exact coverage even at N = 10
    assert abs(_coverage(inference, 10.0, alpha) - (1 - alpha)) < 4 * np.sqrt(alpha * (1 - alpha) / 20000)
  _assert_p_values_consistent_with_intervals(inference, 10.0)


def test_t_test_with_baseline_comparison_is_the_paired_test():
  rng = np.random.default_rng(1)
  baseline = rng.normal(size=(10, 5000))
  main = baseline + rng.normal(size=(10, 5000)) * 0.5
  metrics, baseline_state = _mean_setup(baseline, ('samples', 'replicates'))
  _, main_state = _mean_setup(main, ('samples', 'replicates'))
  inference = t_test.IID.for_baseline_comparison(metrics=metrics, aggregated_statistics=main_state,
                                                 baseline_aggregated_statistics=baseline_state, experimental_unit_dim='samples')
  ref = scipy.stats.ttest_rel(main, baseline, axis=0)
  np.testing.assert_allclose(np.asarray(inference.p_values(0.0)['mean']['variable'].values), ref.pvalue, rtol=1e-8, atol=1e-15)
  np.testing.assert_allclose(np.asarray(inference.point_estimates()['mean']['variable'].values), (main - baseline).mean(axis=0),
                             rtol=1e-9, atol=1e-12)
  for alpha in (0.2, 0.05):                                             # t_test_test.py:56-100
    assert abs(_coverage(inference, 0.0, alpha) - (1 - alpha)) < 4 * np.sqrt(alpha * (1 - alpha) / 5000)
  _assert_p_values_consistent_with_intervals(inference, 0.0)
  # the pieces: renamed statistics on both sides, one state holding both
  comparison = baseline_comparison.BaselineComparison(deterministic.RMSE())
  assert set(comparison.statistics) == {'main_SquaredError', 'baseline_SquaredError'}
  assert comparison.statistics['main_SquaredError'].unique_name == 'main_SquaredError'
  both = baseline_comparison.combine_aggregation_states(main_state, baseline_state)
  assert set(both.sum_weights) == {'main_MeanPrediction', 'baseline_MeanPrediction'}
  assert set(baseline_comparison.for_metrics({'a': MeanPrediction(), 'b': MeanPrediction()}, {'b': MeanPrediction()})) == {'b'}


@pytest.mark.parametrize('make,sample_size,rtol', [
    (t_test.GeerAR2Corrected, 100, 0.75), (t_test.LazarusHACEWC, 100, 0.35),
    (lambda **kw: t_test.LazarusHACEWC(v_0=0.27, **kw), 100, 0.15),
    (t_test.GeerAR2Corrected, 1000, 0.05), (t_test.LazarusHACEWC, 1000, 0.06)])
def test_t_tests_under_autocorrelation(make, sample_size, rtol):
  """t_test_test.py:102-221 with 8000 instead of 50000 replicates (and the tolerances at N = 1000 widened accordingly): AR(2)
  data whose autocorrelation costs a factor ~4.3 in effective sample size."""
  rng = np.random.default_rng(0)
  replicates = 8000
  data = simulate_ar2(rng, mean=10.0, sigma=0.1, phi1=0.5, phi2=0.1, steps=sample_size, replicates=replicates)
  metrics, state = _mean_setup(data, ('steps', 'replicates'))
  inference = make(metrics=metrics, aggregated_statistics=state, experimental_unit_dim='steps')
  for alpha in (0.2, 0.1, 0.05):
    miss = 1 - _coverage(inference, 10.0, alpha)
    slack = 3 * np.sqrt(alpha * (1 - alpha) / replicates)
    assert alpha * (1 - rtol) - slack <= miss <= alpha * (1 + rtol) + slack, (alpha, miss)
  _assert_p_values_consistent_with_intervals(inference, 10.0)
  iid = t_test.IID(metrics=metrics, aggregated_statistics=state, experimental_unit_dim='steps')
  assert 1 - _coverage(iid, 10.0, 0.05) > 0.25                          # the plain test is badly over-confident on this data


def test_ar2_inflation_factor():
  phi1, phi2 = xr.DataArray(np.array(0.5)), xr.DataArray(np.array(0.1))
  k = t_test._inflation_factor_from_ar2_coeffs(phi1, phi2)  # pylint: disable=protected-access
  rho1, rho2 = 0.5 / 0.9, 0.1 + 0.25 / 0.9
  np.testing.assert_allclose(float(k.values) ** 2, (1 - rho1 * 0.5 - rho2 * 0.1) / 0.4 ** 2)
  assert 4.2 < float(k.values) ** 2 < 4.4                               # "around 4.3" (t_test_test.py:193)
  k2 = t_test._inflation_factor_from_ar2_autocorrelation(xr.DataArray(np.array(rho1)), xr.DataArray(np.array(rho2)))  # pylint: disable=protected-access
  np.testing.assert_allclose(float(k2.values), float(k.values), rtol=1e-12)
  # white noise: no inflation; the estimators recover the autocorrelations of a long AR(2) series
  np.testing.assert_allclose(float(t_test._inflation_factor_from_ar2_autocorrelation(  # pylint: disable=protected-access
      xr.DataArray(np.array(0.0)), xr.DataArray(np.array(0.0))).values), 1.0)
  series = simulate_ar2(np.random.default_rng(3), 0.0, 1.0, 0.5, 0.1, steps=200000, replicates=1)[:, 0]
  dev = xr.DataArray(series - series.mean(), dims=('t',), coords={'t': np.arange(series.size)})
  np.testing.assert_allclose(float(t_test._autocorrelation_estimate_from_deviations(dev, 't', 1).values), rho1, atol=0.01)  # pylint: disable=protected-access
  np.testing.assert_allclose(float(t_test._autocorrelation_estimate_from_deviations(dev, 't', 2).values), rho2, atol=0.01)  # pylint: disable=protected-access


def test_cosine_estimator_against_its_definition():
  rng = np.random.default_rng(4)
  n = 60
  data = rng.normal(size=(n, 3)).cumsum(axis=0) * 0.1 + rng.normal(size=(n, 3))
  metrics, state = _mean_setup(data, ('steps', 'series'))
  inference = t_test.LazarusHACEWC(metrics=metrics, aggregated_statistics=state, experimental_unit_dim='steps')
  v = int(0.4 * n ** (2 / 3))
  dev = data - data.mean(axis=0)
  t = np.arange(n) + 0.5
  lam = [np.sqrt(2 / n) * (dev * np.cos(np.pi * j * t / n)[:, None]).sum(axis=0) for j in range(1, v + 1)]   # EWC projections
  want = np.sqrt(np.mean(np.square(lam), axis=0) / n)
  np.testing.assert_allclose(np.asarray(inference.standard_error_estimates()['mean']['variable'].values), want, rtol=1e-10)
  lower, upper = inference.confidence_intervals(0.05)
  q = scipy.stats.t(df=v).ppf(0.975)
  np.testing.assert_allclose(np.asarray(upper['mean']['variable'].values) - np.asarray(lower['mean']['variable'].values), 2 * q * want, rtol=1e-10)
  uneven = xr.DataArray(data[:, 0], dims=('steps',), coords={'steps': np.r_[np.arange(n - 1), n + 5]})
  with pytest.raises(ValueError, match='Non-uniform timestep'):
    t_test.LazarusHACEWC(metrics={'mean': MeanPrediction()}, aggregated_statistics=_state({'MeanPrediction': uneven}),
                         experimental_unit_dim='steps')


def test_t_test_for_constant_sequence():
  """t_test_test.py:223-250: zero-width interval, not NaN."""
  metrics, state = _mean_setup(np.ones(100), ('steps',))
  for cls in (t_test.GeerAR2Corrected, t_test.IID, t_test.LazarusHACEWC):
    inference = cls(metrics=metrics, aggregated_statistics=state, experimental_unit_dim='steps')
    np.testing.assert_allclose(np.asarray(inference.point_estimates()['mean']['variable'].values), 1.0)
    np.testing.assert_allclose(np.asarray(inference.standard_error_estimates()['mean']['variable'].values), 0.0)
    lower, upper = inference.confidence_intervals(alpha=0.05)
    np.testing.assert_allclose(np.asarray(lower['mean']['variable'].values), 1.0)
    np.testing.assert_allclose(np.asarray(upper['mean']['variable'].values), 1.0)
    np.testing.assert_allclose(np.asarray(inference.p_values(null_value=1.0)['mean']['variable'].values), 1.0)
    np.testing.assert_allclose(np.asarray(inference.p_values(null_value=2.0)['mean']['variable'].values), 0.0)


# ---- the linearisation ----------------------------------------------------------------------------------------------------------------
def test_linearised_per_unit_values_against_analytic_gradients():
  rng = np.random.default_rng(5)
  n = 40
  unit = {'init_time': np.arange(n)}
  num = xr.DataArray(rng.normal(size=(n, 3)) + 5.0, dims=('init_time', 'level'), coords=unit)
  den = xr.DataArray(rng.normal(size=(n, 3)) * 0.3 + 2.0, dims=('init_time', 'level'), coords=unit)
  w = xr.DataArray(rng.uniform(0.5, 1.5, size=(n, 3)), dims=('init_time', 'level'), coords=unit)
  state = _state({'MeanPrediction': num, 'MeanTarget': den}, weights=w)
  value, tangents = autodiff.per_unit_values_linearized_around_mean_statistics({'ratio': RatioOfMeans()}, state, 'init_time')
  a, b, ww = num.values * w.values, den.values * w.values, w.values     # per-unit accumulators: two numerators, one denominator each
  ma, mb, mw = a.mean(0), b.mean(0), ww.mean(0)
  np.testing.assert_allclose(np.asarray(value['ratio']['variable'].values), ma / mb, rtol=1e-12)   # (ma / mw) / (mb / mw)
  # f = (A / Wa) / (B / Wb) with Wa = Wb = W here: df = dA / mb - ma dB / mb**2 (the two dW terms cancel)
  want = (a - ma) / mb - ma * (b - mb) / mb ** 2
  got = tangents['ratio']['variable']
  assert got.dims == ('level', 'init_time')
  np.testing.assert_allclose(np.asarray(got.values).T, want, rtol=1e-9, atol=1e-11)
  np.testing.assert_allclose(np.asarray(got.values).mean(axis=1), 0.0, atol=1e-9)     # zero mean along the units
  # a weighted mean alone is already non-linear in its accumulators: d(A / W) = dA / mw - ma dW / mw**2
  value, tangents = autodiff.per_unit_values_linearized_around_mean_statistics({'mean': MeanPrediction()}, _state({'MeanPrediction': num}, weights=w),
                                                                               'init_time')
  np.testing.assert_allclose(np.asarray(tangents['mean']['variable'].values).T, (a - ma) / mw - ma * (ww - mw) / mw ** 2, rtol=1e-9, atol=1e-11)
  # RMSE: sqrt of a mean
  se = xr.DataArray(rng.gamma(2.0, size=(n, 3)), dims=('init_time', 'level'), coords=unit)
  value, tangents = autodiff.per_unit_values_linearized_around_mean_statistics({'rmse': deterministic.RMSE()}, _state({'SquaredError': se}), 'init_time')
  m = se.values.mean(0)
  np.testing.assert_allclose(np.asarray(value['rmse']['variable'].values), np.sqrt(m), rtol=1e-12)
  np.testing.assert_allclose(np.asarray(tangents['rmse']['variable'].values).T, (se.values - m) / (2 * np.sqrt(m)), rtol=1e-9, atol=1e-12)
  with pytest.raises(ValueError, match='No experimental unit coordinate'):
    autodiff.per_unit_values_linearized_around_mean_statistics({'rmse': deterministic.RMSE()}, _state({'SquaredError': se}), 'time')


def test_delta_method_interval_for_a_ratio_of_means():
  rng = np.random.default_rng(6)
  n, replicates = 200, 4000
  num = xr.DataArray(rng.normal(size=(n, replicates)) + 4.0, dims=('units', 'replicates'), coords={'units': np.arange(n)})
  den = xr.DataArray(rng.normal(size=(n, replicates)) * 0.5 + 2.0, dims=('units', 'replicates'), coords={'units': np.arange(n)})
  inference = t_test.IID(metrics={'ratio': RatioOfMeans()}, aggregated_statistics=_state({'MeanPrediction': num, 'MeanTarget': den}),
                         experimental_unit_dim='units')
  for alpha in (0.1, 0.05):
    assert abs(_coverage(inference, 2.0, alpha, name='ratio') - (1 - alpha)) < 4 * np.sqrt(alpha * (1 - alpha) / replicates) + 0.005


def test_utils():
  da = xr.DataArray(np.arange(6.0).reshape(2, 3), dims=('a', 'b'), coords={'a': [10, 20]})
  out = utils.apply_to_slices(lambda x: x * 2, da, dim='a')
  np.testing.assert_array_equal(np.asarray(out.values), da.values * 2)
  out = utils.apply_to_slices(lambda x, y: (x + y).sum('b'), da, da, dim=['a'])
  np.testing.assert_array_equal(np.asarray(out.values), 2 * da.values.sum(axis=1))
  out = utils.apply_to_slices(lambda x: x.isel(a=0, b=0, drop=True).expand_dims(['a', 'b']) + 1, da, dim=('a', 'b'))
  np.testing.assert_array_equal(np.asarray(out.values), da.values + 1)
  with pytest.raises(ValueError, match='not found in any arguments'):
    utils.apply_to_slices(lambda x: x, da, dim='c')
  np.testing.assert_allclose(utils.logarithmic_round(np.array([1.0, 1.04, 9.7, 123.0]), 30), 10 ** (np.round(np.log10([1.0, 1.04, 9.7, 123.0]) * 30) / 30))
  state = _state({'s': da})
  assert utils.get_and_check_experimental_unit_coord(state, 'a').values.tolist() == [10, 20]
  with pytest.raises(ValueError, match='No experimental unit coordinate'):
    utils.get_and_check_experimental_unit_coord(state, 'b')


def test_inference_on_accumulators_from_the_device(backend):
  """The route SURVEY section 8 (f-4) names: area-weighted squared errors reduced over (latitude, longitude) on the device with
  init_time kept, then a paired test of one model against another on those per-init accumulators."""
  del backend
  rng = np.random.default_rng(7)
  n, nlat, nlon = 30, 9, 12
  cs = {'init_time': np.datetime64('2020-01-01', 'ns') + np.arange(n) * np.timedelta64(12, 'h'), 'latitude': np.linspace(-80, 80, nlat),
        'longitude': np.arange(nlon) * 30.0}
  dims = ('init_time', 'latitude', 'longitude')
  truth = rng.normal(size=(n, nlat, nlon)).astype(np.float32)
  shared = rng.normal(size=(n, 1, 1)).astype(np.float32) * 0.5          # errors both models share on a given day
  models = {'good': truth + shared + 0.8 * rng.normal(size=truth.shape).astype(np.float32),
            'poor': truth + shared + 1.0 * rng.normal(size=truth.shape).astype(np.float32)}
  metrics = {'rmse': deterministic.RMSE()}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  states = {}
  for name, field in models.items():
    stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'t': xr.DataArray(field, dims=dims, coords=cs)},
                                                                  {'t': xr.DataArray(truth, dims=dims, coords=cs)})
    states[name] = agg.aggregate_statistics(stats)
  single = t_test.IID(metrics=metrics, aggregated_statistics=states['good'], experimental_unit_dim='init_time')
  per_init = states['good'].mean_statistics()['SquaredError']['t']
  np.testing.assert_allclose(float(single.point_estimates()['rmse']['t'].values), np.sqrt(np.asarray(per_init.values).mean()), rtol=1e-6)
  paired = t_test.LazarusHACEWC.for_baseline_comparison(metrics=metrics, aggregated_statistics=states['good'],
                                                        baseline_aggregated_statistics=states['poor'], experimental_unit_dim='init_time')
  diff = float(paired.point_estimates()['rmse']['t'].values)
  assert diff < 0 and bool(paired.significance_tests(0.0, 0.05)['rmse']['t'].values)   # 'good' has the smaller RMSE, significantly
  lower, upper = paired.confidence_intervals(0.05)
  assert float(lower['rmse']['t'].values) < diff < float(upper['rmse']['t'].values) < 0


# ---- bootstraps (statistical_inference/bootstrap_test.py:26-391, scaled down) ---------------------------------------------------------
from weatherbenchx_amd.statistical_inference import bootstrap  # pylint: disable=g-import-not-at-top,wrong-import-position


class ExpMean(metrics_base.PerVariableMetric):

  @property
  def statistics(self):
    return {'mean_prediction': MeanPrediction()}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return statistic_values['mean_prediction']._unary(np.exp, 'exp')  # pylint: disable=protected-access


def simulate_ar1(rng, mean, sigma_marginal, phi, steps, replicates):
  sigma = sigma_marginal * np.sqrt(1 - phi ** 2)
  out = [sigma_marginal * rng.normal(size=replicates)]
  for _ in range(steps - 1):
    out.append(phi * out[-1] + rng.normal(size=replicates) * sigma)
  return np.stack(out) + mean


def ar1_true_stderr_of_sample_mean(sigma_marginal, phi, n):
  correction = 1 + 2 * phi / (1 - phi) * (1 - (1 - phi ** n) / (1 - phi) / n)
  return sigma_marginal / np.sqrt(n / correction)


def test_iid_bootstrap_inference_of_exp_of_mean():
  rng = np.random.default_rng(0)
  n, datasets, sigma = 100, 1500, 2.0
  sampling = scipy.stats.lognorm(s=sigma / np.sqrt(n), scale=1.0)       # the law of exp(mean of n N(0, sigma) draws)
  data = xr.DataArray(rng.normal(scale=sigma, size=(n, datasets)), dims=('samples', 'replicates'), coords={'samples': np.arange(n)})
  inference = bootstrap.IIDBootstrap(metrics={'exp_mean': ExpMean()}, aggregated_statistics=_state({'MeanPrediction': data}),
                                     experimental_unit_dim='samples', n_replicates=600, rng=rng)
  resampled = inference.resampled_values['exp_mean']['variable']
  assert set(resampled.dims) == {'bootstrap_replicate', 'replicates'} and resampled.sizes['bootstrap_replicate'] == 600
  point = np.asarray(inference.point_estimates()['exp_mean']['variable'].values)
  np.testing.assert_allclose(point, np.exp(data.values.mean(axis=0)), rtol=1e-12)
  np.testing.assert_allclose(point.mean(), sampling.mean(), rtol=0.03)
  stderr = np.asarray(inference.standard_error_estimates()['exp_mean']['variable'].values)
  np.testing.assert_allclose(np.sqrt((stderr ** 2).mean()), sampling.std(), rtol=0.08)
  for alpha in (0.05, 0.2):
    miss = 1 - _coverage(inference, sampling.mean(), alpha, name='exp_mean')
    assert alpha * 0.6 - 3 * np.sqrt(alpha / datasets) <= miss <= alpha * 1.5 + 3 * np.sqrt(alpha / datasets), (alpha, miss)
  # p-values against intervals: the same empirical distribution read two ways (up to the interpolation between order statistics)
  p = np.asarray(inference.p_values(sampling.mean())['exp_mean']['variable'].values)
  lower, upper = inference.confidence_intervals(0.1)
  inside = (np.asarray(lower['exp_mean']['variable'].values) <= sampling.mean()) & (sampling.mean() <= np.asarray(upper['exp_mean']['variable'].values))
  assert ((p > 0.1) == inside).mean() > 0.99


def test_cluster_bootstrap_with_equal_values_in_each_cluster():
  rng = np.random.default_rng(1)
  effective, repeat, datasets = 100, 2, 300
  original = rng.normal(size=(effective, datasets))
  ids = np.repeat(rng.choice(effective * 10, size=effective, replace=False), repeat)
  data = xr.DataArray(np.repeat(original, repeat, axis=0), dims=('samples', 'replicates'),
                      coords={'samples': np.arange(effective * repeat), 'cluster': (('samples',), ids)})
  inference = bootstrap.ClusterBootstrap(metrics={'mean': MeanPrediction()}, aggregated_statistics=_state({'MeanPrediction': data}),
                                         experimental_unit_coord='cluster', n_replicates=2000, rng=rng)
  stderr = np.asarray(inference.standard_error_estimates()['mean']['variable'].values)
  np.testing.assert_allclose(np.sqrt((stderr ** 2).mean()), 1 / np.sqrt(effective), rtol=0.03)   # the duplicates add nothing
  naive = bootstrap.IIDBootstrap(metrics={'mean': MeanPrediction()}, aggregated_statistics=_state({'MeanPrediction': data}),
                                 experimental_unit_dim='samples', n_replicates=500, rng=rng)
  naive_stderr = np.asarray(naive.standard_error_estimates()['mean']['variable'].values)
  np.testing.assert_allclose(np.sqrt((naive_stderr ** 2).mean()), 1 / np.sqrt(effective * repeat), rtol=0.05)  # ... which the i.i.d. one misses


def test_stationary_bootstrap_paths_and_block_length():
  np.random.seed(0)
  paths = bootstrap.stationary_bootstrap_indices(n_data=500, mean_block_length=10.0, n_replicates=400)
  assert paths.shape == (500, 400) and paths.min() >= 0 and paths.max() < 500
  continues = (np.diff(paths, axis=0) % 500) == 1
  np.testing.assert_allclose(1 - continues.mean(), 0.1, atol=0.01)     # a block ends with probability 1 / mean length
  assert (bootstrap.stationary_bootstrap_indices(50, 1e9, 3, rng=np.random.default_rng(1))[1:] ==
          (bootstrap.stationary_bootstrap_indices(50, 1e9, 3, rng=np.random.default_rng(1))[:-1] + 1) % 50).all()
  rng = np.random.default_rng(2)
  assert bootstrap.optimal_block_length(rng.normal(size=5000)) < 3.0   # white noise needs no blocks
  assert bootstrap.optimal_block_length(np.ones(100)) == 1.0
  n = 20000
  for phi in (0.5, 0.9):                                                # AR(1): b = (2 phi / (1 - phi**2))**(2/3) N**(1/3)
    got = np.mean([bootstrap.optimal_block_length(simulate_ar1(rng, 0.0, 1.0, phi, n, 1)[:, 0]) for _ in range(6)])
    want = (2 * phi / (1 - phi ** 2)) ** (2 / 3) * n ** (1 / 3)
    assert 0.7 * want < got < 1.3 * want, (phi, got, want)
  assert bootstrap.optimal_block_length(simulate_ar1(rng, 0.0, 1.0, 0.999, 90, 1)[:, 0]) <= np.ceil(min(3 * np.sqrt(90), 30))   # the cap


def test_stationary_bootstrap_inference_of_means_of_ar1_processes():
  """bootstrap_test.py:153-293: the standard error of the mean of AR(1) series, a different block length for each output point."""
  rng = np.random.default_rng(3)
  n, datasets = 3000, 40
  high = simulate_ar1(rng, -10.0, 1.0, 0.9, n, datasets)
  low = simulate_ar1(rng, 10.0, 1.0, 0.2, n, datasets)
  data = xr.DataArray(np.stack([high, low], axis=-1), dims=('steps', 'replicates', 'autocorr'),
                      coords={'steps': np.arange(n), 'autocorr': ['high', 'low']})
  inference = bootstrap.StationaryBootstrap(metrics={'mean': MeanPrediction()}, aggregated_statistics=_state({'MeanPrediction': data}),
                                            experimental_unit_dim='steps', n_replicates=200, rng=rng)
  point = inference.point_estimates()['mean']['variable']
  np.testing.assert_allclose(np.asarray(point.transpose('replicates', 'autocorr').values), data.values.mean(axis=0), rtol=1e-12)
  stderr = inference.standard_error_estimates()['mean']['variable'].transpose('replicates', 'autocorr')
  rms = np.sqrt((np.asarray(stderr.values) ** 2).mean(axis=0))
  np.testing.assert_allclose(rms[0], ar1_true_stderr_of_sample_mean(1.0, 0.9, n), rtol=0.15)
  np.testing.assert_allclose(rms[1], ar1_true_stderr_of_sample_mean(1.0, 0.2, n), rtol=0.10)
  assert rms[0] > 2.5 * rms[1]
  fixed = bootstrap.StationaryBootstrap(metrics={'mean': MeanPrediction()}, aggregated_statistics=_state({'MeanPrediction': data.isel(replicates=[0])}),
                                        experimental_unit_dim='steps', n_replicates=50, mean_block_length=1.0, rng=rng)
  iid_like = np.asarray(fixed.standard_error_estimates()['mean']['variable'].values).ravel()
  np.testing.assert_allclose(iid_like, data.values[:, 0].std(axis=0) / np.sqrt(n), rtol=0.35)   # blocks of one: the i.i.d. answer
  short = _state({'MeanPrediction': data.isel(steps=slice(0, 5), replicates=[0])})
  with pytest.raises(ValueError, match='at least 8 data points'):
    bootstrap.StationaryBootstrap(metrics={'mean': MeanPrediction()}, aggregated_statistics=short, experimental_unit_dim='steps', n_replicates=5)


def test_stationary_bootstrap_of_a_nonlinear_function_of_two_means():
  """bootstrap_test.py:296-388: ratio of the means of two autocorrelated series."""
  rng = np.random.default_rng(4)
  n = 1000

  def setup(datasets):
    num = xr.DataArray(simulate_ar1(rng, 2.0, 6.6, 0.3, n, datasets), dims=('steps', 'replicates'), coords={'steps': np.arange(n)})
    den = xr.DataArray(simulate_ar1(rng, 1.0, 3.3, 0.3, n, datasets) * 0.1 + 4.0, dims=('steps', 'replicates'), coords={'steps': np.arange(n)})
    return {'ratio_of_means': RatioOfMeans()}, _state({'MeanPrediction': num, 'MeanTarget': den})

  metrics, big = setup(4000)
  draws = np.asarray(metrics_base.compute_metrics_from_statistics(metrics, big.sum_along_dims(['steps']).mean_statistics())[
      'ratio_of_means']['variable'].values)
  metrics, state = setup(60)
  inference = bootstrap.StationaryBootstrap(metrics=metrics, aggregated_statistics=state, experimental_unit_dim='steps', n_replicates=200, rng=rng)
  stderr = np.asarray(inference.standard_error_estimates()['ratio_of_means']['variable'].values)
  np.testing.assert_allclose(np.sqrt((stderr ** 2).mean()), draws.std(ddof=1), rtol=0.2)
  assert 0.8 <= _coverage(inference, draws.mean(), 0.1, name='ratio_of_means') <= 0.99
