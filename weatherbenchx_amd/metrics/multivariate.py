"""Multivariate and distribution scores of ensembles, and the relative economic value of probability forecasts (counterpart of
weatherbenchX/metrics/probabilistic.py:339-603, 785-833, 1006-1303, 1346-1527).

None of these is on the path SURVEY section 8 names and none has a kernel of its own: the per-point values are labelled-array
arithmetic on whatever holds the payload (NumPy on the host, torch where a chunk is resident in HBM), and their weighted, binned,
masked means are the Aggregator's reduction -- the same kernels as every other statistic.  `metrics.probabilistic` hands these
names on, so `probabilistic.EnergyScore` etc. resolve as in the reference.
"""
from __future__ import annotations

from typing import Mapping

import numpy as np

from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base
from weatherbenchx_amd.metrics import categorical
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import wrappers
from weatherbenchx_amd.metrics.probabilistic import ENSEMBLE_DIM
from weatherbenchx_amd.metrics.probabilistic import UnbiasedEnsembleMeanSquaredError


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


# ---- relative economic value (probabilistic.py:1006-1303) ---------------------------------------------------------------------------
def _select_optimal_thresholds(values: xr.DataArray, optimal_thresholds: xr.DataArray, method: str | None = None) -> xr.DataArray:
  """`values` at, for every cost/loss ratio (and every other dim the two share, e.g. lead_time), the probability threshold chosen
  for it: `values.sel(threshold=optimal_thresholds, method=method)` without the `threshold` coordinate
  (probabilistic.py:1006-1059).  method None = exact labels (KeyError otherwise), 'nearest' = the closest available one."""
  values, optimal_thresholds = xr.as_dataarray(values), xr.as_dataarray(optimal_thresholds)
  available = np.asarray(values.coords['threshold'].values, dtype=np.float64)
  wanted = np.asarray(optimal_thresholds.values, dtype=np.float64)
  position = np.abs(wanted[..., None] - available).argmin(axis=-1)
  if method is None:
    missing = available[position] != wanted
    if missing.any():
      raise KeyError(f'thresholds {np.unique(wanted[missing])[:5]} not among the available {available}')
  elif method != 'nearest':
    raise ValueError(f'unsupported selection method {method!r}')
  index = xr.DataArray(position, dims=optimal_thresholds.dims,
                       coords={d: optimal_thresholds.coords[d] for d in optimal_thresholds.dims if d in optimal_thresholds.coords})
  plain = values.drop_vars(['threshold'])
  table, index = xr.broadcast(plain, index)              # both [..., threshold, ..., cost_loss_ratio] in one dim order
  axis = table.dims.index('threshold')
  picked = np.take_along_axis(np.asarray(table.values), np.take(np.asarray(index.values), [0], axis=axis), axis=axis)
  out_dims = tuple(d for d in table.dims if d != 'threshold')
  coords = {k: v for k, v in table._coords.items() if 'threshold' not in v[0]}  # pylint: disable=protected-access
  return xr.DataArray._assemble(np.squeeze(picked, axis=axis), out_dims, coords, name=values.name)  # pylint: disable=protected-access


class RelativeEconomicValue(base.Metric):
  """Relative economic value of probability forecasts of a binary event for users with cost/loss ratio C/L who act when the
  forecast probability exceeds a threshold: (E_climate - E_forecast) / (E_climate - E_perfect) with expenses per unit loss
  E_forecast = C/L (TP + FP) + FN,  E_perfect = C/L (TP + FN),  E_climate = min(C/L, TP + FN).

  Predictions are probabilities in [0, 1], targets binary.  The contingency table is accumulated for every probability
  threshold -- by default the N thresholds between the N + 1 possible values of an N-member ensemble, (k + 0.5) / N -- plus the
  two constant decisions (threshold 0: always act; 1: never).  Without `optimal_thresholds` the result has dims
  (threshold, cost_loss_ratio); with them, one threshold is picked per cost/loss ratio (and per any other shared dim) and
  `threshold` is gone.  Default cost/loss ratios: 50 values from 0.005 towards 1 on a log scale.  probabilistic.py:1062-1303."""

  def __init__(self, *, ensemble_size: int | None = None, probability_thresholds: np.ndarray | None = None,
               cost_loss_ratios: np.ndarray | None = None, optimal_thresholds=None,
               optimal_thresholds_select_nearest: bool = False, statistic_suffix: str | None = None):
    if ensemble_size is None and probability_thresholds is None:
      raise ValueError('Either ensemble_size or probability_thresholds must be specified.')
    if probability_thresholds is not None and ensemble_size is not None:
      raise ValueError('Only one of ensemble_size or probability_thresholds must be specified.')
    if probability_thresholds is not None and statistic_suffix is None:
      raise ValueError('If probability_thresholds is specified, statistic_suffix must be specified.')
    ratios = np.geomspace(0.005, 1, 51)[:-1] if cost_loss_ratios is None else np.asarray(cost_loss_ratios)
    self._cost_loss_ratio = xr.DataArray(ratios, dims=['cost_loss_ratio'], coords={'cost_loss_ratio': ratios})
    if probability_thresholds is None:
      self._thresholds = (np.arange(ensemble_size) + 0.5) / ensemble_size
      if statistic_suffix is None:
        statistic_suffix = 'all_thresholds_for_ensemble_size'
    else:
      self._thresholds = np.asarray(probability_thresholds)
    if not (np.all(self._thresholds >= 0.0) and np.all(self._thresholds <= 1.0)):
      raise ValueError(f'Probability thresholds must be in [0, 1], got {self._thresholds=}.')
    self._unique_name_suffix = statistic_suffix or ''
    if optimal_thresholds is not None:
      for var in (optimal_thresholds.values() if isinstance(optimal_thresholds, Mapping) else [optimal_thresholds]):
        var = xr.as_dataarray(var)
        if 'cost_loss_ratio' not in var.dims:
          raise ValueError('optimal_thresholds must have "cost_loss_ratio" dimensions.')
        if ('cost_loss_ratio' not in var.coords
            or not np.array_equal(np.asarray(var.coords['cost_loss_ratio'].values), ratios)):
          raise ValueError('optimal_thresholds must have cost_loss_ratio coordinates with the same values as the '
                           'cost_loss_ratios argument.')
    self._optimal_thresholds = optimal_thresholds
    self._optimal_thresholds_select_nearest = optimal_thresholds_select_nearest

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    binarize = wrappers.ContinuousToBinary(which='predictions', threshold_value=self._thresholds, threshold_dim='threshold',
                                           unique_name_suffix=self._unique_name_suffix)
    return {name: wrappers.WrappedStatistic(getattr(categorical, name)(), binarize)
            for name in ('TruePositives', 'TrueNegatives', 'FalsePositives', 'FalseNegatives')}

  def _with_constant_decisions(self, tp, fp, fn):
    """Threshold 0 (always act: TP = base rate, FP = 1 - base rate, FN = 0) in front, threshold 1 (never act: FN = base rate) behind."""
    base_rate = tp.isel(threshold=0, drop=True) + fn.isel(threshold=0, drop=True)
    zero = base_rate * 0.0
    at = lambda x, threshold: x.expand_dims(threshold=[threshold])
    return (xr.concat([at(base_rate, 0.0), tp, at(zero, 1.0)], dim='threshold'),
            xr.concat([at(1.0 - base_rate, 0.0), fp, at(zero, 1.0)], dim='threshold'),
            xr.concat([at(zero, 0.0), fn, at(base_rate, 1.0)], dim='threshold'))

  def values_from_mean_statistics(self, statistic_values):
    names = list(self.statistics)
    common = set.intersection(*[set(statistic_values[s]) for s in names])
    return {var: self._values_from_mean_statistics_per_variable({s: statistic_values[s][var] for s in names}, var)
            for var in statistic_values[names[0]] if var in common}

  def _values_from_mean_statistics_per_variable(self, statistic_values, var_name):
    def ordered(da):  # the statistics carry `threshold` wherever the transform put it; the table below wants it first
      da = xr.as_dataarray(da)
      return da.transpose('threshold', *[d for d in da.dims if d != 'threshold'])
    tp, fp, fn = self._with_constant_decisions(*(ordered(statistic_values[s])
                                                 for s in ('TruePositives', 'FalsePositives', 'FalseNegatives')))
    if self._optimal_thresholds is not None:
      chosen = self._optimal_thresholds[var_name] if isinstance(self._optimal_thresholds, Mapping) else self._optimal_thresholds
      method = 'nearest' if self._optimal_thresholds_select_nearest else None
      tp, fp, fn = (_select_optimal_thresholds(x, chosen, method) for x in (tp, fp, fn))
    ratio = self._cost_loss_ratio
    forecast = ratio * (tp + fp) + fn
    perfect = ratio * (tp + fn)
    climate = np.minimum(ratio, tp + fn)
    return (climate - forecast) / (climate - perfect)
