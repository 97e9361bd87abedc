"""Bootstrap confidence intervals and tests: i.i.d., cluster and stationary (block) resampling of the experimental units
(counterpart of weatherbenchX/statistical_inference/bootstrap.py:35-555).

A replicate is a reweighting of the per-unit accumulators: the metric of  sum_i c_i (weighted statistics)_i / sum_i c_i (weights)_i
with resampling counts c.  For the i.i.d. and cluster bootstraps all replicates are one contraction of the accumulators with a
[replicate, unit] count matrix (`AggregationState.dot`); the stationary bootstrap draws its index paths per output series, each
with its own block length.  The interval is the percentile interval of the replicates; a bootstrap targets the expectation of the
finite-sample estimator, which for a non-linear metric is not quite the metric of the expectations.

Block length selection: the reference calls `arch.bootstrap.optimal_block_length` (third-party `arch`, not in the reference tree and
not installed here).  `optimal_block_length` below restates the published procedure it implements -- Politis & White (2004),
"Automatic block-length selection for the dependent bootstrap", Econometric Reviews 23:53-70, with the correction of Patton,
Politis & White (2009), Econometric Reviews 28:372-375 -- and is pinned by its own known answers (white noise -> 1, AR(1) -> the
closed form); agreement with `arch` to the last digit is NOT checked (parity unpinned for this one number, which only sets how
efficiently the blocks capture the dependence).

Random numbers come from `numpy.random`'s global state, as in the reference (seed it with `numpy.random.seed`), or from the
`rng=` Generator given to a constructor."""
from __future__ import annotations

import functools
from typing import Hashable, Mapping, final

import numpy as np

from weatherbenchx_amd import aggregation
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.statistical_inference import autodiff
from weatherbenchx_amd.statistical_inference import base
from weatherbenchx_amd.statistical_inference import utils

_REPLICATE_DIM = 'bootstrap_replicate'


def _over_replicates(fn):
  """`fn(values [..., replicate]) -> [...]` applied to a DataArray that has the replicate dim."""
  def apply(da):
    da = xr.as_dataarray(da)
    moved = da.transpose(*[d for d in da.dims if d != _REPLICATE_DIM], _REPLICATE_DIM)
    values = np.asarray(moved.values, dtype=np.float64)
    import warnings  # pylint: disable=g-import-not-at-top
    with warnings.catch_warnings(), np.errstate(all='ignore'):
      warnings.simplefilter('ignore', RuntimeWarning)                   # (all-NaN series)
      out = fn(values)
    return moved.isel({_REPLICATE_DIM: 0}, drop=True)._replace(data=np.asarray(out))  # pylint: disable=protected-access
  return apply


class Bootstrap(base.StatisticalInferenceMethod):
  """What every resampling scheme shares once the replicates exist (bootstrap.py:35-118)."""

  _resampled_values: base.MetricValues
  _point_estimates: base.MetricValues

  @property
  def resampled_values(self) -> base.MetricValues:
    """The metric on every resampled dataset, along `bootstrap_replicate`."""
    return self._resampled_values

  def point_estimates(self) -> base.MetricValues:
    return self._point_estimates

  @final
  def standard_error_estimates(self) -> base.MetricValues:
    return xarray_tree.map_structure(_over_replicates(lambda v: np.nanstd(v, axis=-1, ddof=1)), self.resampled_values)

  @final
  def confidence_intervals(self, alpha: float = 0.05):
    at = lambda q: xarray_tree.map_structure(_over_replicates(lambda v: np.nanquantile(v, q, axis=-1)), self.resampled_values)
    return at(alpha / 2), at(1 - alpha / 2)

  @final
  def p_values(self, null_value: float = 0.) -> base.MetricValues:
    """Two-sided: twice the smaller tail of the replicates' empirical distribution (linear between order statistics) at the null."""
    def one(series: np.ndarray) -> float:
      series = np.sort(series[~np.isnan(series)])
      if series.size == 0:
        return np.nan
      cdf = np.interp(null_value, series, np.linspace(0, 1, series.size))
      return 2 * min(cdf, 1 - cdf)
    return xarray_tree.map_structure(_over_replicates(lambda v: np.apply_along_axis(one, -1, v)), self.resampled_values)


def _multinomial_counts(rng, n_units: int, n_replicates: int) -> np.ndarray:
  source = np.random if rng is None else rng
  return source.multinomial(n_units, np.full(n_units, 1 / n_units), size=n_replicates)


def _metrics_of(metrics, state: aggregation.AggregationState):
  return metrics_base.compute_metrics_from_statistics(metrics, state.mean_statistics())


class IIDBootstrap(Bootstrap):
  """Units resampled independently with replacement (bootstrap.py:121-147)."""

  def __init__(self, metrics: Mapping[str, metrics_base.Metric], aggregated_statistics: aggregation.AggregationState,
               experimental_unit_dim: str, n_replicates: int, rng: np.random.Generator | None = None):
    n = utils.get_and_check_experimental_unit_coord(aggregated_statistics, experimental_unit_dim).size
    counts = xr.DataArray(_multinomial_counts(rng, n, n_replicates).astype(np.float64), dims=[_REPLICATE_DIM, experimental_unit_dim])
    self._point_estimates = _metrics_of(metrics, aggregated_statistics.sum_along_dims([experimental_unit_dim]))
    self._resampled_values = _metrics_of(metrics, aggregated_statistics.dot(counts, dim=experimental_unit_dim))


class ClusterBootstrap(Bootstrap):
  """Whole clusters -- the units that share a value of a 1-D coordinate, which need not be an index -- resampled independently:
  arbitrary dependence inside a cluster, none between clusters (Davison & Hinkley 1997, strategy 1; bootstrap.py:150-226).  Clusters
  may differ in size; a resampled dataset then has the original number of clusters, not of units."""

  def __init__(self, metrics, aggregated_statistics: aggregation.AggregationState, experimental_unit_coord: str, n_replicates: int,
               rng: np.random.Generator | None = None):
    coord = utils.get_and_check_experimental_unit_coord(aggregated_statistics, experimental_unit_coord, check_is_dim=False)
    unit_dim = coord.dims[0]
    clusters, member_of = np.unique(np.asarray(coord.values), return_inverse=True)
    counts = _multinomial_counts(rng, clusters.size, n_replicates)[:, member_of]      # every unit takes its cluster's count
    counts = xr.DataArray(counts.astype(np.float64), dims=[_REPLICATE_DIM, unit_dim])
    self._point_estimates = _metrics_of(metrics, aggregated_statistics.sum_along_dims([unit_dim]))
    self._resampled_values = _metrics_of(metrics, aggregated_statistics.dot(counts, dim=unit_dim))


def stationary_bootstrap_indices(n_data: int, mean_block_length: float, n_replicates: int, dtype=np.int64, rng=None) -> np.ndarray:
  """Index paths [n_data, n_replicates] of the stationary bootstrap: start anywhere, then at every step continue to the next
  (cyclic) position, or with probability 1 / mean_block_length jump to a fresh random one -- blocks of geometric length
  (bootstrap.py:229-247)."""
  source = np.random if rng is None else rng
  draw = (lambda: source.randint(n_data, size=(n_replicates,), dtype=dtype)) if rng is None else (
      lambda: source.integers(n_data, size=(n_replicates,), dtype=dtype))
  uniform = (lambda: source.rand(n_replicates)) if rng is None else (lambda: source.random(n_replicates))
  jump = 1 / mean_block_length
  paths = np.empty((n_data, n_replicates), dtype=dtype)
  paths[0] = draw()
  for step in range(1, n_data):
    fresh_block = uniform() < jump
    fresh = draw()
    paths[step] = np.where(fresh_block, fresh, (paths[step - 1] + 1) % n_data)
  return paths


def optimal_block_length(series: np.ndarray) -> float:
  """Mean block length of the stationary bootstrap that minimises the mean squared error of its variance estimate for the mean of
  `series` (Politis & White 2004 with the 2009 correction):
      b = (2 G**2 / D)**(1/3) N**(1/3),   G = sum_k lambda(k / M) |k| R(k),   D = 2 (sum_k lambda(k / M) R(k))**2,
  R the sample autocovariances, lambda the trapezoidal flat-top window (1 up to 1/2, then linear to 0 at 1), M = 2 m with m the
  first lag after which K_N = max(5, sqrt(log10 N)) consecutive autocorrelations stay inside +-2 sqrt(log10 N / N); b is capped at
  min(3 sqrt(N), N / 3)."""
  x = np.asarray(series, dtype=np.float64).ravel()
  n = x.size
  d = x - x.mean()
  r0 = float(d @ d) / n
  if r0 == 0:
    return 1.0
  k_n = max(5, int(np.ceil(np.sqrt(np.log10(n)))))
  m_max = int(np.ceil(np.sqrt(n))) + k_n
  m_max = min(m_max, n - 1)
  acov = np.array([float(d[k:] @ d[:n - k]) / n for k in range(m_max + 1)])
  inside = np.abs(acov / r0) < 2 * np.sqrt(np.log10(n) / n)
  m_hat = None
  for m in range(0, m_max - k_n + 1):
    if inside[m + 1:m + k_n + 1].all():
      m_hat = m
      break
  big_m = m_max if m_hat is None else min(2 * max(m_hat, 1), m_max)
  k = np.arange(1, big_m + 1)
  window = np.where(k / big_m <= 0.5, 1.0, 2 * (1 - k / big_m))
  g = 2 * float((window * k * acov[1:big_m + 1]).sum())
  spectrum0 = acov[0] + 2 * float((window * acov[1:big_m + 1]).sum())
  if spectrum0 == 0:
    return 1.0
  b = (2 * g ** 2 / (2 * spectrum0 ** 2)) ** (1 / 3) * n ** (1 / 3)
  return float(min(b, np.ceil(min(3 * np.sqrt(n), n / 3))))


class StationaryBootstrap(Bootstrap):
  """Stationary (geometric-block) bootstrap of Politis & Romano (1994) for serially dependent units (bootstrap.py:250-555).  Not
  limited to means: every replicate re-evaluates the metric on resampled accumulators (weights resampled with their statistics).
  The block length is chosen per output series -- per metric, variable and point of any extra dims -- from that series' linearised
  per-unit values (`autodiff`; exact for linear metrics), unless `mean_block_length` fixes it.  Assumes stationarity: strong
  seasonality or trends in the scores should be removed first."""

  def __init__(self, metrics: Mapping[str, metrics_base.Metric], aggregated_statistics: aggregation.AggregationState,
               experimental_unit_dim: str, n_replicates: int, mean_block_length: float | None = None,
               block_length_rounding_resolution: float | None = 30.0, stationary_bootstrap_indices_cache_size: int = 50,
               rng: np.random.Generator | None = None):
    self._unit = experimental_unit_dim
    self._mean_block_length = mean_block_length
    self._n_replicates = n_replicates
    self._state = aggregated_statistics
    self._rounding = block_length_rounding_resolution
    # (index paths are reused for series whose rounded block lengths agree)
    self._paths = functools.lru_cache(maxsize=stationary_bootstrap_indices_cache_size)(
        functools.partial(stationary_bootstrap_indices, rng=rng))
    self._point_estimates, tangents = autodiff.per_unit_values_linearized_around_mean_statistics(
        metrics, aggregated_statistics, experimental_unit_dim)
    self._resampled_values = {name: self._for_metric(metric, self._point_estimates[name], tangents[name])
                              for name, metric in metrics.items()}

  def _block_length(self, series: xr.DataArray) -> float:
    if self._mean_block_length is not None:
      return self._mean_block_length
    if series.sizes[self._unit] < 8:
      raise ValueError(f'Need at least 8 data points along experimental_unit_dim {self._unit} to set mean_block_length '
                       'automatically -- and many more than 8 recommended.')
    values = np.asarray(series.values).squeeze()
    assert values.ndim == 1
    length = max(1.0, optimal_block_length(values))
    return float(utils.logarithmic_round(length, self._rounding)) if self._rounding is not None else length

  def _for_metric(self, metric, point_estimates: Mapping[Hashable, xr.DataArray], tangents: Mapping[Hashable, xr.DataArray]):
    pick = lambda field: {name: getattr(self._state, field)[stat.unique_name] for name, stat in metric.statistics.items()}
    sws, sw = pick('sum_weighted_statistics'), pick('sum_weights')
    out = {}
    for var in point_estimates:
      if len(point_estimates) > 1 and all(var in per_var for per_var in sws.values()):
        mine = lambda tree, var=var: {name: {var: per_var[var]} for name, per_var in tree.items()}   # the usual per-variable metric
        sws_var, sw_var = mine(sws), mine(sw)
      else:
        sws_var, sw_var = sws, sw
      dims = tuple(xr.as_dataarray(point_estimates[var]).dims)
      one = functools.partial(self._for_series, metric, var)
      out[var] = utils.apply_to_slices(one, tangents[var], sws_var, sw_var, dim=dims) if dims else one(tangents[var], sws_var, sw_var)
    return out

  def _for_series(self, metric, var, tangents: xr.DataArray, sws, sw) -> xr.DataArray:
    n = tangents.sizes[self._unit]
    paths = self._paths(n_data=n, mean_block_length=self._block_length(tangents), n_replicates=self._n_replicates)

    def resampled_sum(da):
      da = xr.as_dataarray(da)
      moved = da.transpose(*[d for d in da.dims if d != self._unit], self._unit)
      total = np.asarray(moved.values, dtype=np.float64)[..., paths].sum(axis=-2)      # [..., unit] -> [..., unit, replicate] -> sum
      frame = moved.isel({self._unit: 0}, drop=True)
      return xr.DataArray._assemble(total, frame.dims + (_REPLICATE_DIM,), dict(frame._coords), name=da.name)  # pylint: disable=protected-access

    sums, weights = xarray_tree.map_structure(resampled_sum, (sws, sw))
    means = xarray_tree.map_structure(lambda a, b: a / b, sums, weights)
    return metric.values_from_mean_statistics(means)[var]
