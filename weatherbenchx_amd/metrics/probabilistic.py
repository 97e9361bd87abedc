"""Ensemble statistics and metrics (counterpart of the CRPS / spread-skill part of
weatherbenchX/metrics/probabilistic.py:28-336, 606-688, 864-1003).

Per-point arithmetic lives in csrc/wbx_ens_impl.hpp (one lane owns one grid point's M members in
VGPRs: sorting-network rank form for `use_sort=True`, pairwise form for `use_sort=False`).
`skipna_ensemble=True` and float64 / M > 64 members run on the generic (memory re-reading, fp64 pair form) kernel;
targets that carry the ensemble dim are handled member by member (CRPSSkill, UnbiasedEnsembleMeanSquaredError,
`which='targets'`); their combination with skipna_ensemble=True (per-point counts on both sides: not linear in the
target member) is evaluated un-fused on the labeled arrays and reduced through the generic (PASS1) kernels.
"""
from __future__ import annotations

from typing import Mapping

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import lazy
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import wrappers

ENSEMBLE_DIM = 'number'


def _sqrt(da):
  return np.sqrt(da)


class EnsembleAveragedStatistic(base.Statistic):
  """Mean of a wrapped statistic over the ensemble dim (probabilistic.py:35-69)."""

  def __init__(self, wrapped_statistic: base.Statistic, *, ensemble_dim: str, skipna_ensemble: bool):
    self._wrapped_statistic = wrapped_statistic
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def unique_name(self) -> str:
    return self._wrapped_statistic.unique_name + '_each_' + self._ensemble_dim

  def compute(self, predictions, targets):
    out = {}
    for name, da in self._wrapped_statistic.compute(predictions, targets).items():
      if self._ensemble_dim not in da.dims:
        raise ValueError(f'Dimension {self._ensemble_dim} not found in {da.dims}')
      out[name] = da.mean(dim=self._ensemble_dim, skipna=self._skipna_ensemble)
    return out


class EnsembleAveragedMetric(base.Metric):
  """Any metric with its statistics averaged over the ensemble dim (probabilistic.py:72-113)."""

  def __init__(self, wrapped_metric: base.Metric, *, ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False):
    self._wrapped_metric = wrapped_metric
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {name: EnsembleAveragedStatistic(stat, ensemble_dim=self._ensemble_dim,
                                            skipna_ensemble=self._skipna_ensemble)
            for name, stat in self._wrapped_metric.statistics.items()}

  def values_from_mean_statistics(self, statistic_values):
    return self._wrapped_metric.values_from_mean_statistics(statistic_values)


class CRPSSkill(base.PerVariableStatistic):
  """E|X - Y| (probabilistic.py:116-145)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def unique_name(self) -> str:
    return f'CRPSSkill_{self._ensemble_dim}'

  def _compute_per_variable(self, predictions, targets):
    if self._ensemble_dim in targets.dims:
      # mean over both ensemble dims of |p_i - t_j| (probabilistic.py:134-145) = mean over target members of the
      # per-member skill: linear, so one fused launch per target member
      if self._skipna_ensemble:
        # NaN members on either side: the mean runs over the non-NaN (prediction member, target member) pairs of each
        # point, which is not linear in the target member any more: the pair kernel with per-point counts (wbx_ens2_partial)
        return lazy.ens2_statistic('CRPSSkill', predictions, targets, self._ensemble_dim, skipna_ensemble=True)
      members = lazy.target_members(targets, self._ensemble_dim)
      terms = [lazy.ens_statistic('CRPSSkill', predictions, tj, self._ensemble_dim) for tj in members]
      return lazy.LinearCombination(terms, scale=1.0 / len(terms), name=predictions.name)
    return lazy.ens_statistic('CRPSSkill', predictions, targets, self._ensemble_dim,
                              skipna_ensemble=self._skipna_ensemble)


class CRPSSpread(base.PerVariableStatistic):
  """E|X - X'| estimate, fair or not, rank or pairwise form (probabilistic.py:165-247).
  As in the reference, `unique_name` ignores use_sort / skipna_ensemble (probabilistic.py:189-192)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, use_sort: bool = False, fair: bool = True,
               which: str = 'predictions', skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._use_sort = use_sort
    self._which = which
    self._fair = fair
    self._skipna_ensemble = skipna_ensemble

  @property
  def unique_name(self) -> str:
    return f"CRPSSpread_{self._ensemble_dim}_{'fair' if self._fair else 'unfair'}_{self._which}"

  def _compute_per_variable(self, predictions, targets):
    if self._which == 'predictions':
      da = predictions
    elif self._which == 'targets':
      da = targets
    else:
      raise ValueError(f'Unhandled {self._which=}')
    if self._ensemble_dim not in da.dims:
      raise ValueError(f'Dimension {self._ensemble_dim} not found in {da.dims}')
    if not self._skipna_ensemble and da.sizes[self._ensemble_dim] < 2:
      raise ValueError('Cannot estimate CRPS spread with n_ensemble < 2.')
    if self._skipna_ensemble and self._use_sort:
      raise ValueError('skipna_ensemble is not supported with use_sort=True.')
    # the spread only looks at `da`; any member-free companion field serves as the kernel's target operand
    other = targets if self._which == 'predictions' else predictions
    if self._ensemble_dim in other.dims or self._which == 'targets':
      other = lazy.target_members(da, self._ensemble_dim)[0]
    return lazy.ens_statistic('CRPSSpread', da, other, self._ensemble_dim, use_sort=self._use_sort, fair=self._fair,
                              skipna_ensemble=self._skipna_ensemble, member_only=True)


class EnsembleVariance(base.PerVariableStatistic):
  """Unbiased (ddof=1) variance over the ensemble dim (probabilistic.py:250-273)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def unique_name(self) -> str:
    return f'EnsembleVariance_{self._ensemble_dim}_skipna_ensemble_{self._skipna_ensemble}'

  def _compute_per_variable(self, predictions, targets):
    other = targets if self._ensemble_dim not in targets.dims else lazy.target_members(targets, self._ensemble_dim)[0]
    return lazy.ens_statistic('EnsembleVariance', predictions, other, self._ensemble_dim,
                              skipna_ensemble=self._skipna_ensemble, member_only=True)


class UnbiasedEnsembleMeanSquaredError(base.PerVariableStatistic):
  """(mean_m X - Y)**2 - var_m(X)/M (probabilistic.py:276-336)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def unique_name(self) -> str:
    return f'UnbiasedEnsembleMeanSquaredError_{self._ensemble_dim}_skipna_ensemble_{self._skipna_ensemble}'

  def _compute_per_variable(self, predictions, targets):
    if self._ensemble_dim not in predictions.dims:
      raise ValueError(f'Dimension {self._ensemble_dim} not found in {predictions.dims}')
    if self._ensemble_dim in targets.dims:
      # (mean p - mean t)^2 - var_p / M - var_t / N (probabilistic.py:320-336).  With
      #   mean_j (a - t_j)^2 = (a - mean t)^2 + (N - 1) / N * var_t      (var_t with ddof = 1)
      # this is  mean_j UEMSE(p, t_j) - var_t : N fused launches against single target members plus the variance
      # lane of one launch over the target ensemble -- linear, so the accumulators are combined after the reduction.
      if self._skipna_ensemble:
        # per-point member counts on both sides (probabilistic.py:304-333): lane 1 of the same launch (wbx_ens2_partial)
        return lazy.ens2_statistic('UnbiasedEnsembleMeanSquaredError', predictions, targets, self._ensemble_dim, skipna_ensemble=True)
      members = lazy.target_members(targets, self._ensemble_dim)
      n = len(members)
      if n < 2:
        raise ValueError('an ensemble of targets needs at least 2 members (its variance has ddof=1)')
      terms = [lazy.ens_statistic('UnbiasedEnsembleMeanSquaredError', predictions, tj, self._ensemble_dim)
               for tj in members]
      terms.append(lazy.ens_statistic('EnsembleVariance', targets, members[0], self._ensemble_dim, member_only=True))
      return lazy.LinearCombination(terms, coeffs=[1.0 / n] * n + [-1.0], name=predictions.name)
    return lazy.ens_statistic('UnbiasedEnsembleMeanSquaredError', predictions, targets, self._ensemble_dim,
                              skipna_ensemble=self._skipna_ensemble)


class CRPSEnsemble(base.PerVariableMetric):
  """CRPS = E|X - Y| - 0.5 E|X - X'| (probabilistic.py:606-688)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, use_sort: bool = False, fair: bool = True,
               skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._use_sort = use_sort
    self._fair = fair
    self._skipna_ensemble = skipna_ensemble

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {
        'CRPSSkill': CRPSSkill(ensemble_dim=self._ensemble_dim, skipna_ensemble=self._skipna_ensemble),
        'CRPSSpread': CRPSSpread(ensemble_dim=self._ensemble_dim, use_sort=self._use_sort, fair=self._fair,
                                 skipna_ensemble=self._skipna_ensemble),
    }

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return statistic_values['CRPSSkill'] - 0.5 * statistic_values['CRPSSpread']


class EnsembleErrorExceedance(deterministic.ErrorExceedance):
  """Error exceedance averaged over the ensemble members, NaN members skipped (probabilistic.py:836-861): per
  threshold the fraction of members whose absolute error exceeds it -- counted in registers while the members
  stream by once (csrc/wbx_cat.hip)."""

  def __init__(self, thresholds, ensemble_dim: str = ENSEMBLE_DIM):
    super().__init__(thresholds=thresholds)
    self._ensemble_dim = ensemble_dim

  def _compute_per_variable(self, predictions, targets):
    values, dim, coord = deterministic._threshold_array(self._thresholds, predictions.name)  # pylint: disable=protected-access
    return lazy.cat_statistic(_hip.CAT_EXCEED, predictions, targets, dim, coord, thresholds=values,
                              ensemble_dim=self._ensemble_dim)


class RankHistogram(base.PerVariableStatistic):
  """One-hot of the target's rank among the M members along a new `rank` dimension of M + 1 bins
  (probabilistic.py:1306-1343)."""

  def __init__(self, *, ensemble_dim: str = ENSEMBLE_DIM):
    self._ensemble_dim = ensemble_dim

  @property
  def unique_name(self) -> str:
    return f'RankHistogram_{self._ensemble_dim}'

  def _compute_per_variable(self, predictions, targets):
    if self._ensemble_dim not in predictions.dims:
      raise ValueError(f'Dimension {self._ensemble_dim} not found in {predictions.dims}')
    m = predictions.sizes[self._ensemble_dim]
    return lazy.cat_statistic(_hip.CAT_RANK, predictions, targets, 'rank', np.arange(m + 1),
                              ensemble_dim=self._ensemble_dim)


class CRPSEnsembleDistance(base.PerVariableMetric):
  """E|X - Y| - 0.5 E|X - X'| - 0.5 E|Y - Y'| for ensemble-valued predictions AND targets
  (probabilistic.py:691-782)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, use_sort: bool = False, fair: bool = True,
               skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._use_sort = use_sort
    self._fair = fair
    self._skipna_ensemble = skipna_ensemble

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {
        'CRPSSkill': CRPSSkill(ensemble_dim=self._ensemble_dim),
        'CRPSSpread': CRPSSpread(ensemble_dim=self._ensemble_dim, use_sort=self._use_sort, fair=self._fair,
                                 skipna_ensemble=self._skipna_ensemble),
        'CRPSTargetSpread': CRPSSpread(ensemble_dim=self._ensemble_dim, use_sort=self._use_sort, fair=self._fair,
                                       which='targets'),
    }

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return (statistic_values['CRPSSkill'] - 0.5 * statistic_values['CRPSSpread']
            - 0.5 * statistic_values['CRPSTargetSpread'])


class UnbiasedEnsembleMeanRMSE(base.PerVariableMetric):
  """sqrt(mean UnbiasedEnsembleMeanSquaredError) (probabilistic.py:864-894)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {'UnbiasedEnsembleMeanSquaredError': UnbiasedEnsembleMeanSquaredError(
        ensemble_dim=self._ensemble_dim, skipna_ensemble=self._skipna_ensemble)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return _sqrt(statistic_values['UnbiasedEnsembleMeanSquaredError'])


def SpreadSkillRatio(**unused_kwargs):  # pylint: disable=invalid-name
  # Same behaviour as the reference: the class was withdrawn (probabilistic.py:897-902).
  raise ValueError('SpreadSkillRatio is no longer supported as it was not correctly implemented. '
                   'Please use UnbiasedSpreadSkillRatio instead and see the docstring of that class for more details.')


class UnbiasedSpreadSkillRatio(base.PerVariableMetric):
  """sqrt(mean EnsembleVariance / mean UnbiasedEnsembleMeanSquaredError) (probabilistic.py:905-967)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    kw = dict(ensemble_dim=self._ensemble_dim, skipna_ensemble=self._skipna_ensemble)
    return {'EnsembleVariance': EnsembleVariance(**kw),
            'UnbiasedEnsembleMeanSquaredError': UnbiasedEnsembleMeanSquaredError(**kw)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return _sqrt(statistic_values['EnsembleVariance'] / statistic_values['UnbiasedEnsembleMeanSquaredError'])


class EnsembleRootMeanVariance(base.PerVariableMetric):
  """sqrt(mean EnsembleVariance) (probabilistic.py:970-1003)."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False):
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {'EnsembleVariance': EnsembleVariance(ensemble_dim=self._ensemble_dim,
                                                 skipna_ensemble=self._skipna_ensemble)}

  def _values_from_mean_statistics_per_variable(self, mean_statistic_values):
    return _sqrt(mean_statistic_values['EnsembleVariance'])


# ---- multivariate and distribution scores (probabilistic.py:339-603, 785-833, 1346-1527) --------------------------------------
# Per-point labelled-array arithmetic on whatever holds the payload (NumPy on the host, torch where a chunk is resident); the
# weighted, binned reduction of the result is the Aggregator's, i.e. the same kernels as every other statistic.

def _dims_tuple(dim) -> tuple:
  return (dim,) if isinstance(dim, str) else tuple(dim)


def _norm_over(da: xr.DataArray, dims: tuple) -> xr.DataArray:
  """Euclidean norm over `dims`; a NaN anywhere in the vector makes its norm NaN."""
  return (da * da).sum(dims, skipna=False)._unary(np.sqrt, 'sqrt')  # pylint: disable=protected-access


class EnsembleRankedProbabilityScore(base.PerVariableStatistic):
  """RPS of an ensemble against bin thresholds: the squared error between the prediction's and the target's empirical CDF at
  every threshold, summed over `bin_dim`.  `fair=True` uses the unbiased ensemble-mean squared error per threshold (debiasing
  in the ensemble size, for whichever side is an ensemble); otherwise the squared error of the ensemble-mean CDFs
  (probabilistic.py:339-477)."""

  def __init__(self, prediction_bin_thresholds, target_bin_thresholds, bin_dim: str, unique_name_suffix: str,
               ensemble_dim: str = ENSEMBLE_DIM, skipna_ensemble: bool = False, fair: bool = True,
               enforce_monotonicity: bool = True, right_inclusive: bool = True):
    self._ensemble_dim = ensemble_dim
    self._skipna_ensemble = skipna_ensemble
    self._fair = fair
    self._bin_dim = bin_dim
    self._unique_name_suffix = unique_name_suffix
    cdf = {which: wrappers.ContinuousToCDF(which=which, threshold_values=values, threshold_dim=bin_dim,
                                           unique_name_suffix=unique_name_suffix, enforce_monotonicity=enforce_monotonicity,
                                           right_inclusive=right_inclusive)
           for which, values in (('predictions', prediction_bin_thresholds), ('targets', target_bin_thresholds))}
    if fair:
      per_threshold = UnbiasedEnsembleMeanSquaredError(ensemble_dim=ensemble_dim, skipna_ensemble=skipna_ensemble)
    else:
      # (targets without the ensemble dim are left as they are; an ensemble of targets is averaged too)
      per_threshold = wrappers.WrappedStatistic(
          deterministic.SquaredError(),
          wrappers.EnsembleMean(which='both', ensemble_dim=ensemble_dim, skipna=skipna_ensemble,
                                skip_if_ensemble_dim_missing=True))
    self._per_threshold = wrappers.WrappedStatistic(wrappers.WrappedStatistic(per_threshold, cdf['targets']), cdf['predictions'])

  @property
  def unique_name(self) -> str:
    return (f'RankedProbabilityScore_{self._ensemble_dim}_skipna_ensemble_{self._skipna_ensemble}_fair_{self._fair}_'
            f'{self._unique_name_suffix}')

  def _compute_per_variable(self, predictions, targets):
    squared = self._per_threshold.compute({'_': predictions}, {'_': targets})['_']
    return xr.as_dataarray(squared).sum(self._bin_dim, skipna=self._skipna_ensemble)


class EnergyScoreSkill(base.PerVariableStatistic):
  """mean_m ||X_m - Y|| with the norm over `dim` (probabilistic.py:480-503)."""

  def __init__(self, dim, ensemble_dim: str = 'sample'):
    self._dim = dim
    self._ensemble_dim = ensemble_dim

  @property
  def unique_name(self):
    return f'EnergyScore_Skill_dim={self._dim}_ensemble_dim={self._ensemble_dim}'

  def _compute_per_variable(self, predictions, targets):
    return _norm_over(predictions - targets, _dims_tuple(self._dim)).mean(self._ensemble_dim, skipna=False)


class EnergyScoreSpread(base.PerVariableStatistic):
  """sum_{m, m'} ||X_m - X_m'|| / (M (M - 1))  (M**2 when not fair), the norm over `dim` (probabilistic.py:506-551).

  The M x M table of the reference is never formed: one pass per offset k = 1 .. M - 1 pairs every member with the one k
  places on (cyclically), which visits each ordered pair (m, m' != m) once; the diagonal contributes nothing."""

  def __init__(self, dim, ensemble_dim: str = 'sample', fair: bool = True):
    self._dim = dim
    self._ensemble_dim = ensemble_dim
    self._fair = fair

  @property
  def unique_name(self):
    return f'EnergyScore_Spread_dim={self._dim}_ensemble_dim={self._ensemble_dim}_fair={self._fair}'

  def _compute_per_variable(self, predictions, targets):
    del targets
    e = self._ensemble_dim
    m = predictions.sizes[e]
    plain = predictions.drop_vars([e]) if e in predictions.coords else predictions
    total = None
    for k in range(1, m):
      rolled = plain.isel({e: np.roll(np.arange(m), -k)})
      part = _norm_over(plain - rolled, _dims_tuple(self._dim)).sum(e, skipna=False)
      total = part if total is None else total + part
    if total is None:  # one member: no pairs; 0 / 0 for the fair estimate, like the reference's empty sum over the divider
      total = _norm_over(plain - plain, _dims_tuple(self._dim)).sum(e, skipna=False)
    with np.errstate(all='ignore'):
      return total / float(m * (m - 1) if self._fair else m * m)


class VariogramScore(base.PerVariableStatistic):
  """sum_{i, j} (|y_i - y_j|**p - mean_m |x_i^m - x_j^m|**p)**2 over all pairs along `dim` (Scheuerer & Hamill 2015;
  probabilistic.py:554-603).  Same pairing by cyclic offsets as EnergyScoreSpread: [N, N] tables are never formed."""

  def __init__(self, dim: str, ensemble_dim: str, p: float = 0.5):
    self._dim = dim
    self._ensemble_dim = ensemble_dim
    self._p = p

  @property
  def unique_name(self):
    return f'VariogramScore_dim={self._dim}_ensemble_dim={self._ensemble_dim}'

  def _compute_per_variable(self, predictions, targets):
    d, n = self._dim, predictions.sizes[self._dim]
    strip = lambda a: a.drop_vars([d]) if d in a.coords else a
    x, y = strip(predictions), strip(targets)
    total = None
    for k in range(n):  # (k = 0 is the diagonal: zero unless NaN, which it must hand on)
      order = np.roll(np.arange(n), -k)
      ty = abs(y - y.isel({d: order})) ** self._p
      tx = (abs(x - x.isel({d: order})) ** self._p).mean(self._ensemble_dim, skipna=False)
      part = ((ty - tx) ** 2).sum(d, skipna=False)
      total = part if total is None else total + part
    return total


class WassersteinDistance(base.PerVariableStatistic):
  """1-Wasserstein (earth mover's) distance between the prediction ensemble and the target ensemble at every point; the two
  ensembles may differ in size (probabilistic.py:785-833, there through scipy.stats.wasserstein_distance point by point).

  Here for all points at once: W1 = integral |F_p - F_t|.  The pooled members are sorted, the two empirical CDFs are running
  counts of where each sorted value came from, and the integral is the sum over the gaps between consecutive pooled values."""

  def __init__(self, ensemble_dim: str = ENSEMBLE_DIM):
    self._ensemble_dim = ensemble_dim

  @property
  def unique_name(self) -> str:
    return f'WassersteinDistance_{self._ensemble_dim}'

  def _compute_per_variable(self, predictions, targets):
    e = self._ensemble_dim
    if e not in predictions.dims:
      raise ValueError(f'Ensemble dimension {e!r} not found in predictions: {predictions}')
    if e not in targets.dims:
      raise ValueError(f'Ensemble dimension {e!r} not found in targets: {targets}')
    strip = lambda a: a.drop_vars([e]) if e in a.coords else a
    p, t = xr.broadcast(strip(predictions).rename({e: '_p'}), strip(targets).rename({e: '_t'}))
    frame = tuple(d for d in p.dims if d not in ('_p', '_t'))
    p = p.isel(_t=0).transpose(*frame, '_p')
    t = t.isel(_p=0).transpose(*frame, '_t')
    m, n = p.sizes['_p'], t.sizes['_t']
    pd, td = p.data, t.data
    if xr._is_torch(pd) or xr._is_torch(td):  # pylint: disable=protected-access
      import torch  # pylint: disable=g-import-not-at-top
      dev = pd.device if xr._is_torch(pd) else td.device  # pylint: disable=protected-access
      pd, td = (torch.as_tensor(a, device=dev).to(torch.float64) for a in (pd, td))
      pooled, order = torch.sort(torch.cat([pd, td], dim=-1), dim=-1, stable=True)
      from_p = (order < m).to(torch.float64)
      gap = torch.abs(torch.cumsum(from_p, -1) / m - torch.cumsum(1.0 - from_p, -1) / n)[..., :-1]
      dist = (gap * (pooled[..., 1:] - pooled[..., :-1])).sum(-1)
    else:
      both = np.concatenate([np.asarray(pd, dtype=np.float64), np.asarray(td, dtype=np.float64)], axis=-1)
      order = np.argsort(both, axis=-1, kind='stable')
      pooled = np.take_along_axis(both, order, axis=-1)
      from_p = (order < m).astype(np.float64)
      gap = np.abs(np.cumsum(from_p, -1) / m - np.cumsum(1.0 - from_p, -1) / n)[..., :-1]
      dist = (gap * np.diff(pooled, axis=-1)).sum(-1)
    coords = {k: v for k, v in p._coords.items() if set(v[0]) <= set(frame)}  # pylint: disable=protected-access
    return xr.DataArray._assemble(dist, frame, coords, name=predictions.name)  # pylint: disable=protected-access


class EnergyScore(base.PerVariableMetric):
  """E||X - Y|| - 0.5 E||X - X'|| with the norm over `dim` (Gneiting & Raftery; probabilistic.py:1346-1406)."""

  def __init__(self, dim, ensemble_dim: str = ENSEMBLE_DIM, fair: bool = True):
    self._dim = dim
    self._ensemble_dim = ensemble_dim
    self._fair = fair

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {'EnergyScoreSkill': EnergyScoreSkill(dim=self._dim, ensemble_dim=self._ensemble_dim),
            'EnergyScoreSpread': EnergyScoreSpread(dim=self._dim, ensemble_dim=self._ensemble_dim, fair=self._fair)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return statistic_values['EnergyScoreSkill'] - 0.5 * statistic_values['EnergyScoreSpread']


def _tile(window_size: int, window_dim: str, wrap_longitude: bool):
  return wrappers.Tile(which='both', window_size=window_size, window_dim=window_dim, wrap_longitude=wrap_longitude)


class TiledEnergyScore(base.PerVariableMetric):
  """EnergyScore of every window_size x window_size patch of the grid (rows without a full window at the top and bottom are
  dropped, longitude optionally wraps): probabilistic.py:1409-1467."""

  _WINDOW_DIM = 'window'

  def __init__(self, window_size: int = 3, ensemble_dim: str = ENSEMBLE_DIM, fair: bool = True, wrap_longitude: bool = True):
    self._window_size = window_size
    self._ensemble_dim = ensemble_dim
    self._fair = fair
    self._wrap_longitude = wrap_longitude

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    tile = _tile(self._window_size, self._WINDOW_DIM, self._wrap_longitude)
    return {
        'TiledEnergyScore_Skill': wrappers.WrappedStatistic(
            EnergyScoreSkill(dim=self._WINDOW_DIM, ensemble_dim=self._ensemble_dim), tile),
        'TiledEnergyScore_Spread': wrappers.WrappedStatistic(
            EnergyScoreSpread(dim=self._WINDOW_DIM, ensemble_dim=self._ensemble_dim, fair=self._fair), tile),
    }

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return statistic_values['TiledEnergyScore_Skill'] - 0.5 * statistic_values['TiledEnergyScore_Spread']


class TiledVariogramScore(base.PerVariableMetric):
  """VariogramScore of every window_size x window_size patch: the pairs stay among correlated neighbours
  (probabilistic.py:1470-1527)."""

  _WINDOW_DIM = 'window'

  def __init__(self, window_size: int = 3, ensemble_dim: str = ENSEMBLE_DIM, p: float = 0.5, wrap_longitude: bool = True):
    self._window_size = window_size
    self._ensemble_dim = ensemble_dim
    self._p = p
    self._wrap_longitude = wrap_longitude

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    tile = _tile(self._window_size, self._WINDOW_DIM, self._wrap_longitude)
    return {'TiledVariogramScore': wrappers.WrappedStatistic(
        VariogramScore(dim=self._WINDOW_DIM, ensemble_dim=self._ensemble_dim, p=self._p), tile)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return statistic_values['TiledVariogramScore']
