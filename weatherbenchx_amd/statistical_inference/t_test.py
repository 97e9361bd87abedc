"""t-tests for (smooth functions of) means over experimental units, plain and robust to autocorrelation (counterpart of
weatherbenchX/statistical_inference/t_test.py:36-485).

All three share the frame: the metric at the mean accumulators, per-unit first-order deviations from it (`autodiff`, the delta
method), a standard error estimated from those deviations, a Student t distribution with some degrees of freedom.  They differ in
the standard error:
  IID               s / sqrt(N), N - 1 degrees of freedom -- units independent;
  GeerAR2Corrected  the same inflated by the factor k of a stationary AR(2) process fitted through the lag-1 / lag-2
                    autocorrelations (Geer 2016, Tellus A 68, 30229); degrees of freedom unchanged;
  LazarusHACEWC     long-run variance from the v = v_0 N**(2/3) lowest-frequency cosine projections of the series (equal-weighted
                    cosine estimator, Lazarus, Lewis, Stock & Watson 2018, JBES 36, 541-559), v degrees of freedom -- no
                    parametric model of the dependence; with `for_baseline_comparison` a Diebold-Mariano-type test.
"""
from __future__ import annotations

import abc
import dataclasses
from typing import Mapping, final

import numpy as np
import scipy.fft
import scipy.stats

from weatherbenchx_amd import aggregation
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.statistical_inference import autodiff
from weatherbenchx_amd.statistical_inference import base


def _check_constant(data_array: xr.DataArray, dim: str, error_suffix: str = ''):
  values = np.asarray(data_array.values)
  first = np.take(values, [0], axis=data_array.dims.index(dim))
  same = np.allclose(first, values) if values.dtype.kind == 'f' else bool(np.all(first == values))
  if not same:
    raise ValueError(f'Found non-constant values along dimension {dim} for {data_array.name}. {error_suffix}')


def _check_uniform_step(data_array: xr.DataArray, dim: str) -> None:
  """A numeric coordinate along `dim` has to advance in equal steps (the autocorrelation estimates assume it)."""
  if dim in data_array.coords:
    coord = np.asarray(data_array.coords[dim].values)
    if np.issubdtype(coord.dtype, np.number) and coord.size > 1:
      _check_constant(xr.DataArray(np.diff(coord), dims=(dim,), name=dim), dim, 'Non-uniform timestep not supported.')


def _variance_estimate_from_deviations(deviations: xr.DataArray, dim: str, ddof: int = 1) -> xr.DataArray:
  return (deviations ** 2).sum(dim, skipna=False) / (deviations.sizes[dim] - ddof)


def _autocorrelation_estimate_from_deviations(deviations: xr.DataArray, dim: str, lag: int = 1) -> xr.DataArray:
  """mean of d_t d_(t+lag) over the N - lag pairs, over the (ddof = 1) variance; 0 where the variance is 0."""
  variance = _variance_estimate_from_deviations(deviations, dim)
  bare = deviations.drop_vars([k for k in deviations.coords if dim in deviations.coords[k].dims])
  n = bare.sizes[dim]
  result = (bare.isel({dim: slice(0, n - lag)}) * bare.isel({dim: slice(lag, None)})).mean(dim, skipna=False) / variance
  return result.where(variance != 0, 0)


def _sqrt(da: xr.DataArray) -> xr.DataArray:
  return da._unary(np.sqrt, 'sqrt')  # pylint: disable=protected-access


def _inflation_factor_from_ar2_coeffs(phi1, phi2):
  """sqrt of (variance of the mean of an AR(2) series) / (the same for white noise of equal variance)."""
  rho1 = phi1 / (1 - phi2)
  rho2 = phi2 + phi1 ** 2 / (1 - phi2)
  return _sqrt((1 - rho1 * phi1 - rho2 * phi2) / (1 - phi1 - phi2) ** 2)


def _inflation_factor_from_ar2_autocorrelation(rho1, rho2):
  """Yule-Walker: the AR(2) coefficients from the first two autocorrelations, then the factor above (Geer 2016, eq. 9-13)."""
  den = 1 - rho1 ** 2
  phi1 = rho1 * (1 - rho2) / den
  phi2 = (rho2 - rho1 ** 2) / den
  return _sqrt((1 - rho1 * phi1 - rho2 * phi2) / (1 - phi1 - phi2) ** 2)


@dataclasses.dataclass(frozen=True)
class _TTestResults:
  """One variable of one metric."""
  mean: xr.DataArray
  standard_error: xr.DataArray
  degrees_of_freedom: int

  def _quantile(self, alpha: float) -> float:
    return float(-scipy.stats.t(df=self.degrees_of_freedom).ppf(alpha / 2))

  def ci_lower(self, alpha: float = 0.05) -> xr.DataArray:
    return self.mean - self.standard_error * self._quantile(alpha)

  def ci_upper(self, alpha: float = 0.05) -> xr.DataArray:
    return self.mean + self.standard_error * self._quantile(alpha)

  def p_value(self, null_value: float = 0.) -> xr.DataArray:
    difference = self.mean - null_value
    score = (difference / self.standard_error).where(~((difference == 0) & (self.standard_error == 0)), 0.)
    cdf = scipy.stats.t(df=self.degrees_of_freedom).cdf(np.abs(np.asarray(score.values, dtype=np.float64)))
    return score._replace(data=2 * (1 - cdf))  # pylint: disable=protected-access


class _Base(base.StatisticalInferenceMethod):
  """The shared frame; subclasses supply the standard error and the degrees of freedom (t_test.py:133-237).  The t-test assumes
  roughly Gaussian per-unit values (the central limit theorem helps for larger samples) and, for non-linear metrics, a function
  close to linear over the sampling variation of the mean statistics."""

  def __init__(self, metrics: Mapping[str, metrics_base.Metric], aggregated_statistics: aggregation.AggregationState,
               experimental_unit_dim: str):
    values, tangents = autodiff.per_unit_values_linearized_around_mean_statistics(metrics, aggregated_statistics,
                                                                                  experimental_unit_dim)
    self._results = xarray_tree.map_structure(
        lambda mean, deviations: self._compute_results(experimental_unit_dim, xr.as_dataarray(mean), deviations), values, tangents)

  @abc.abstractmethod
  def _compute_results(self, experimental_unit_dim: str, mean: xr.DataArray, per_unit_deviations: xr.DataArray) -> _TTestResults:
    ...

  def _each(self, fn):
    return xarray_tree.map_structure(fn, self._results)

  @final
  def point_estimates(self):
    return self._each(lambda r: r.mean)

  @final
  def standard_error_estimates(self):
    return self._each(lambda r: r.standard_error)

  @final
  def confidence_intervals(self, alpha: float = 0.05):
    return self._each(lambda r: r.ci_lower(alpha)), self._each(lambda r: r.ci_upper(alpha))

  @final
  def p_values(self, null_value: float = 0.):
    return self._each(lambda r: r.p_value(null_value))


class IID(_Base):
  """The classic one-sample t-test (t_test.py:240-256)."""

  def _compute_results(self, experimental_unit_dim, mean, per_unit_deviations):
    n = per_unit_deviations.sizes[experimental_unit_dim]
    variance = _variance_estimate_from_deviations(per_unit_deviations, experimental_unit_dim, ddof=1)
    return _TTestResults(mean, _sqrt(variance / n), n - 1)


class GeerAR2Corrected(_Base):
  """Standard error inflated for AR(2) autocorrelation between consecutive units (t_test.py:259-310).  Well motivated
  asymptotically when the series is a stationary AR(2) process; optimistic (intervals too narrow) for small samples or strong
  autocorrelation, since neither the noise of the fitted coefficients nor the loss of effective sample size enters the degrees
  of freedom."""

  def _compute_results(self, experimental_unit_dim, mean, per_unit_deviations):
    _check_uniform_step(per_unit_deviations, experimental_unit_dim)
    n = per_unit_deviations.sizes[experimental_unit_dim]
    variance = _variance_estimate_from_deviations(per_unit_deviations, experimental_unit_dim, ddof=1)
    k = _inflation_factor_from_ar2_autocorrelation(
        _autocorrelation_estimate_from_deviations(per_unit_deviations, experimental_unit_dim, lag=1),
        _autocorrelation_estimate_from_deviations(per_unit_deviations, experimental_unit_dim, lag=2))
    return _TTestResults(mean, _sqrt(variance / n) * k, n - 1)


class LazarusHACEWC(_Base):
  """Heteroscedasticity- and autocorrelation-consistent t-test with the equal-weighted cosine estimator (t_test.py:313-485).
  Only the v lowest-frequency cosine components of the series estimate the variance of the mean -- the higher ones are the ones
  autocorrelation distorts -- with v = v_0 N**(2/3) (at least 1, at most N - 1, where the test is the IID one).  v_0 = 0.4, the
  authors' recommendation, keeps the size of a 5 % test accurate up to autocorrelations around 0.7 at some cost in power when the
  dependence is weaker; a larger v_0 buys power for size distortion, a smaller one the opposite (their table 2b)."""

  def __init__(self, metrics, aggregated_statistics, experimental_unit_dim: str, v_0: float = 0.4):
    self._v_0 = v_0
    super().__init__(metrics, aggregated_statistics, experimental_unit_dim)

  def _compute_results(self, experimental_unit_dim, mean, per_unit_deviations):
    _check_uniform_step(per_unit_deviations, experimental_unit_dim)
    n = per_unit_deviations.sizes[experimental_unit_dim]
    v = min(max(1, int(self._v_0 * n ** (2 / 3))), n - 1)
    series = per_unit_deviations.transpose(*[d for d in per_unit_deviations.dims if d != experimental_unit_dim], experimental_unit_dim)
    projections = scipy.fft.dct(np.asarray(series.values, dtype=np.float64), type=2, axis=-1, norm='ortho')[..., 1:v + 1]
    long_run_variance = series.isel({experimental_unit_dim: 0}, drop=True)._replace(data=np.mean(projections ** 2, axis=-1))  # pylint: disable=protected-access
    return _TTestResults(mean=mean, standard_error=_sqrt(long_run_variance / n), degrees_of_freedom=v)
