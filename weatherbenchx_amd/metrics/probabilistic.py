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
from weatherbenchx_amd.metrics import base
from weatherbenchx_amd.metrics import deterministic

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


# ---- the scores without a kernel of their own (labelled-array arithmetic per point, the Aggregator's reduction after it) live in
# metrics/multivariate.py and are reachable under the reference's names here.
_ELSEWHERE = ('EnsembleRankedProbabilityScore', 'EnergyScoreSkill', 'EnergyScoreSpread', 'VariogramScore', 'WassersteinDistance',
              'EnergyScore', 'TiledEnergyScore', 'TiledVariogramScore', 'RelativeEconomicValue', '_select_optimal_thresholds')


def __getattr__(name):
  if name in _ELSEWHERE:
    from weatherbenchx_amd.metrics import multivariate  # pylint: disable=g-import-not-at-top
    return getattr(multivariate, name)
  raise AttributeError(f'module {__name__!r} has no attribute {name!r}')


def __dir__():
  return sorted(list(globals()) + list(_ELSEWHERE))
