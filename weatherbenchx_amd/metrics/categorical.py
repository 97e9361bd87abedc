"""Contingency-table statistics and the scores built on them (counterpart of weatherbenchX/metrics/categorical.py:25-971).

The four per-point indicators (TP / TN / FP / FN of binary predictions and targets) are labelled-array arithmetic on whatever
holds the payload; their weighted, binned, masked means are the Aggregator's reduction (the same kernels as every other
statistic), and every score below is a formula on those four means.  Inputs come from `wrappers.ContinuousToBinary` /
`ContinuousToBins` (thresholded fields, probability bins).

SEEPS and the interval statistics (Confident / Covered / JaccardDistant -> Opportunism) follow the same pattern with a
climatology beside the chunk.
"""
from __future__ import annotations

from typing import Hashable, Mapping, Sequence, Union

import numpy as np

from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base
from weatherbenchx_amd.metrics import wrappers


def _indicator(predictions: xr.DataArray, targets: xr.DataArray, predicted: bool, observed: bool) -> xr.DataArray:
  """float32 1 where (prediction is positive) == predicted and (target is positive) == observed, NaN where either input is NaN
  (categorical.py:36-41: `astype(bool)` products under `.where(~isnan(p * t))`)."""
  p = predictions != 0
  t = targets != 0
  hit = (p if predicted else ~p) & (t if observed else ~t)
  valid = ~(predictions * targets).isnull()
  return hit.astype(np.float32).where(valid)


class _Indicator(base.PerVariableStatistic):
  _predicted: bool
  _observed: bool

  @property
  def unique_name(self) -> str:
    return type(self).__name__

  def _compute_per_variable(self, predictions, targets):
    return _indicator(predictions, targets, self._predicted, self._observed)


class TruePositives(_Indicator):
  """Predicted and observed (categorical.py:25-42)."""
  _predicted, _observed = True, True


class TrueNegatives(_Indicator):
  """Neither predicted nor observed (categorical.py:45-62)."""
  _predicted, _observed = False, False


class FalsePositives(_Indicator):
  """Predicted, not observed (categorical.py:65-82)."""
  _predicted, _observed = True, False


class FalseNegatives(_Indicator):
  """Observed, not predicted (categorical.py:85-101)."""
  _predicted, _observed = False, True


class RankedProbabilityScore(base.PerVariableStatistic):
  """sum over `bin_dim` of (CDF_prediction - CDF_target)**2 for inputs that already ARE cumulative distributions along `bin_dim`
  (categorical.py:307-340; from ensembles: probabilistic.EnsembleRankedProbabilityScore)."""

  def __init__(self, bin_dim: str):
    self._bin_dim = bin_dim

  @property
  def unique_name(self) -> str:
    return 'RankedProbabilityScore'

  def _compute_per_variable(self, predictions, targets):
    return ((predictions - targets) ** 2).sum(self._bin_dim)


_CELLS = {'tp': TruePositives, 'fp': FalsePositives, 'fn': FalseNegatives, 'tn': TrueNegatives}


class _ContingencyScore(base.PerVariableMetric):
  """A score of the mean contingency table: `_needs` names the cells it reads, `_score` is the formula."""
  _needs: Sequence[str] = ()

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {_CELLS[c].__name__: _CELLS[c]() for c in self._needs}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return self._score(**{c: statistic_values[_CELLS[c].__name__] for c in self._needs})

  def _score(self, **cells):
    raise NotImplementedError


class CSI(_ContingencyScore):
  """Critical success index (threat score) TP / (TP + FP + FN).  categorical.py:345-370."""
  _needs = ('tp', 'fp', 'fn')

  def _score(self, tp, fp, fn):
    return tp / (tp + fp + fn)


class Accuracy(_ContingencyScore):
  """(TP + TN) / all.  categorical.py:373-400."""
  _needs = ('tp', 'fp', 'fn', 'tn')

  def _score(self, tp, fp, fn, tn):
    return (tp + tn) / (tp + fp + fn + tn)


class Recall(_ContingencyScore):
  """Hit rate TP / (TP + FN).  categorical.py:403-423."""
  _needs = ('tp', 'fn')

  def _score(self, tp, fn):
    return tp / (tp + fn)


class FalseAlarmRate(_ContingencyScore):
  """FP / (TP + FP) -- the false alarm RATIO of the forecasts, named as in the reference.  categorical.py:426-446."""
  _needs = ('tp', 'fp')

  def _score(self, tp, fp):
    return fp / (tp + fp)


class Precision(_ContingencyScore):
  """TP / (TP + FP).  categorical.py:449-469."""
  _needs = ('tp', 'fp')

  def _score(self, tp, fp):
    return tp / (tp + fp)


class F1Score(_ContingencyScore):
  """2 TP / (2 TP + FP + FN).  categorical.py:472-500."""
  _needs = ('tp', 'fp', 'fn')

  def _score(self, tp, fp, fn):
    return 2 * tp / (2 * tp + fp + fn)


class FrequencyBias(_ContingencyScore):
  """Predicted positives over observed positives (TP + FP) / (TP + FN).  categorical.py:503-524."""
  _needs = ('tp', 'fp', 'fn')

  def _score(self, tp, fp, fn):
    return (tp + fp) / (tp + fn)


class HSS(_ContingencyScore):
  """Heidke skill score 2 (TP TN - FP FN) / ((TP + FN)(FN + TN) + (TP + FP)(FP + TN)).  categorical.py:527-553."""
  _needs = ('tp', 'fp', 'fn', 'tn')

  def _score(self, tp, fp, fn, tn):
    return 2 * (tp * tn - fp * fn) / ((tp + fn) * (fn + tn) + (tp + fp) * (fp + tn))


class ETS(_ContingencyScore):
  """Equitable threat (Gilbert skill) score: CSI with the hits expected by chance, (TP + FP)(TP + FN) / all, taken out of
  numerator and denominator.  categorical.py:556-587."""
  _needs = ('tp', 'fp', 'fn', 'tn')

  def _score(self, tp, fp, fn, tn):
    chance = (tp + fp) * (tp + fn) / (tp + fp + fn + tn)
    return (tp - chance) / (tp + fp + fn - chance)


class SEDI(_ContingencyScore):
  """Symmetric extremal dependence index (Ferro & Stephenson 2011) of the hit rate H = TP / (TP + FN) and the false alarm rate
  F = FP / (FP + TN), both clipped to [1e-6, 1 - 1e-6]:
  (ln F - ln H + ln(1 - H) - ln(1 - F)) / (ln H + ln F + ln(1 - H) + ln(1 - F)).  categorical.py:590-635."""
  _needs = ('tp', 'fp', 'fn', 'tn')

  def _score(self, tp, fp, fn, tn):
    h = (tp / (tp + fn)).clip(1e-6, 1 - 1e-6)
    f = (fp / (fp + tn)).clip(1e-6, 1 - 1e-6)
    ln = lambda a: a._unary(np.log, 'log')  # pylint: disable=protected-access
    return (ln(f) - ln(h) + ln(1 - h) - ln(1 - f)) / (ln(h) + ln(f) + ln(1 - h) + ln(1 - f))


class Reliability(base.PerVariableMetric):
  """Calibration curve: predicted probabilities are put into bins (ten of width 0.1 by default, `ContinuousToBins`), and per bin
  the observed frequency of the event is TP / (TP + FP).  categorical.py:638-698."""

  def __init__(self, bin_values: Sequence[float] = (-np.inf, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.),
               bin_dim: str = 'reliability_bin', statistic_suffix: str | None = None):
    self._bin_values = bin_values
    self._bin_dim = bin_dim
    self._unique_name_suffix = statistic_suffix

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    bins = wrappers.ContinuousToBins(which='predictions', bin_values=self._bin_values, bin_dim=self._bin_dim,
                                     unique_name_suffix=self._unique_name_suffix)
    return {'TruePositives': wrappers.WrappedStatistic(TruePositives(), bins),
            'FalsePositives': wrappers.WrappedStatistic(FalsePositives(), bins)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    tp, fp = statistic_values['TruePositives'], statistic_values['FalsePositives']
    return tp / (tp + fp)


def _ensemble_quantile(da: xr.DataArray, q: float, dim: str) -> xr.DataArray:
  return wrappers.EnsembleQuantiles(which='both', quantiles=[q], ensemble_dim=dim, quantile_dim='_q').transform_fn(da).isel(_q=0, drop=True)


class Covered(base.PerVariableStatistic):
  """Whether the target lies inside the ensemble's [low, high] quantile interval (linear-interpolated quantiles, both ends
  included).  categorical.py:750-785."""

  def __init__(self, ensemble_dim: str, interval_quantile_boundaries: tuple = (0.1, 0.9)):
    self._ensemble_dim = ensemble_dim
    self._interval_low, self._interval_high = interval_quantile_boundaries

  @property
  def unique_name(self) -> str:
    return f'Covered_interval_low={self._interval_low}_interval_high={self._interval_high}'

  def _compute_per_variable(self, predictions, targets):
    low = _ensemble_quantile(predictions, self._interval_low, self._ensemble_dim)
    high = _ensemble_quantile(predictions, self._interval_high, self._ensemble_dim)
    return (low <= targets) & (targets <= high)


def _aligned(climatology) -> xr.DataArray:
  """The climatology at the chunk's valid times as a plain labelled array (the base class hands over an index table that the
  kernels of ACC gather through; the interval statistics below need the values)."""
  return climatology.aligned_view() if hasattr(climatology, 'aligned_view') else xr.as_dataarray(climatology)


class Confident(base.PerVariableStatisticWithClimatology):
  """Whether the ensemble's quantile spread (high - low) is below `confidence_threshold` times the spread of the climatological
  quantiles; the climatology carries a `quantile` dim with those two levels.  categorical.py:701-747."""

  def __init__(self, ensemble_dim: str, climatology, spread_quantile_boundaries: tuple = (0.1, 0.9),
               confidence_threshold: float = 0.7):
    super().__init__(climatology)
    self._ensemble_dim = ensemble_dim
    self._spread_low, self._spread_high = spread_quantile_boundaries
    self._confidence_threshold = confidence_threshold

  @property
  def unique_name(self) -> str:
    return f'Confident_conf_thres={self._confidence_threshold}_spread_low={self._spread_low}_spread_high={self._spread_high}'

  def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
    del targets
    clim = _aligned(aligned_climatology)
    spread = (_ensemble_quantile(predictions, self._spread_high, self._ensemble_dim)
              - _ensemble_quantile(predictions, self._spread_low, self._ensemble_dim))
    clim_spread = clim.sel(quantile=self._spread_high, drop=True) - clim.sel(quantile=self._spread_low, drop=True)
    return spread < self._confidence_threshold * clim_spread


class JaccardDistant(base.PerVariableStatisticWithClimatology):
  """Whether the Jaccard distance 1 - |A n B| / |A u B| between the forecast interval A (ensemble quantiles) and the climatological
  interval B exceeds `threshold`; two identical single-point intervals overlap fully (distance 0).  categorical.py:788-863."""

  def __init__(self, ensemble_dim: str, climatology, threshold: float = 0.75, interval_quantile_boundaries: tuple = (0.1, 0.9)):
    super().__init__(climatology)
    self._ensemble_dim = ensemble_dim
    self._threshold = threshold
    self._interval_low, self._interval_high = interval_quantile_boundaries

  @property
  def unique_name(self) -> str:
    return f'JaccardDistant_threshold={self._threshold}_interval_low={self._interval_low}_interval_high={self._interval_high}'

  def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
    del targets
    clim = _aligned(aligned_climatology)
    a_lo = _ensemble_quantile(predictions, self._interval_low, self._ensemble_dim)
    a_hi = _ensemble_quantile(predictions, self._interval_high, self._ensemble_dim)
    b_lo, b_hi = clim.sel(quantile=self._interval_low, drop=True), clim.sel(quantile=self._interval_high, drop=True)
    overlap = (np.minimum(a_hi, b_hi) - np.maximum(a_lo, b_lo)).clip(min=0)
    union = (a_hi - a_lo) + (b_hi - b_lo) - overlap
    index = (overlap / union).where(union > 0, 1.0)
    return (1 - index) > self._threshold


class Opportunism(base.PerVariableMetric):
  """Product of the mean fractions of forecasts that are (or are not) confident, covered and Jaccard-distant; the last two take
  part only when their flag is given.  categorical.py:866-971."""

  def __init__(self, ensemble_dim: str, climatology, is_confident: bool, is_covered: bool | None = None,
               is_jaccard_distant: bool | None = None, confidence_quantile_boundaries: tuple = (0.1, 0.9),
               coverage_quantile_boundaries: tuple = (0.1, 0.9), jaccard_distance_quantile_boundaries: tuple = (0.1, 0.9),
               confidence_threshold: float = 0.7, jaccard_distance_threshold: float = 0.75):
    self._flags = {'Confident': is_confident, 'Covered': is_covered, 'JaccardDistant': is_jaccard_distant}
    self._ensemble_dim = ensemble_dim
    self._climatology = climatology
    self._boundaries = {'Confident': confidence_quantile_boundaries, 'Covered': coverage_quantile_boundaries,
                        'JaccardDistant': jaccard_distance_quantile_boundaries}
    self._confidence_threshold = confidence_threshold
    self._jaccard_distance_threshold = jaccard_distance_threshold

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    # confidence is always evaluated; the other two only when they are used
    out = {'Confident': Confident(ensemble_dim=self._ensemble_dim, climatology=self._climatology,
                                  spread_quantile_boundaries=self._boundaries['Confident'],
                                  confidence_threshold=self._confidence_threshold)}
    if self._flags['Covered'] is not None:
      out['Covered'] = Covered(ensemble_dim=self._ensemble_dim, interval_quantile_boundaries=self._boundaries['Covered'])
    if self._flags['JaccardDistant'] is not None:
      out['JaccardDistant'] = JaccardDistant(ensemble_dim=self._ensemble_dim, climatology=self._climatology,
                                             threshold=self._jaccard_distance_threshold,
                                             interval_quantile_boundaries=self._boundaries['JaccardDistant'])
    return out

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    out = None
    for name, flag in self._flags.items():
      if name != 'Confident' and flag is None:
        continue
      factor = statistic_values[name] if flag else 1 - statistic_values[name]
      out = factor if out is None else out * factor
    return out


class SEEPS(base.Statistic):
  """Stable equitable error in probability space (Rodwell et al. 2010) of precipitation: forecast and observation are each put
  into dry (<= dry_threshold_mm, in metres here) / light / heavy (>= the climatological wet threshold at the valid time), and the
  pair is charged half the entry of the 3 x 3 matrix below, which depends on the climatological dry fraction p1 of the place:

              observed:   dry                   light          heavy
      forecast dry        0                     1 / (1 - p1)   4 / (1 - p1)
      forecast light      1 / p1                0              3 / (1 - p1)
      forecast heavy      1 / p1 + 3 / (2 + p1) 3 / (2 + p1)   0

  `climatology` holds `<variable>_seeps_dry_fraction` and `<variable>_seeps_threshold` over (dayofyear, hour, latitude,
  longitude).  Places with p1 outside [min_p1, max_p1] are NaN, and the result carries that as a `mask` coordinate (combined with a
  mask the predictions OR the targets already have) for `Aggregator(masked=True)`.  categorical.py:104-304."""

  def __init__(self, variables: Sequence[str], climatology, dry_threshold_mm: Union[float, Sequence[float]] = 0.25,
               min_p1: Union[float, Sequence[float]] = 0.1, max_p1: Union[float, Sequence[float]] = 0.85):
    per_variable = lambda v: list(v) if isinstance(v, Sequence) else [v] * len(variables)
    self._variables = list(variables)
    self._climatology = climatology
    self._dry_threshold_mm, self._min_p1, self._max_p1 = per_variable(dry_threshold_mm), per_variable(min_p1), per_variable(max_p1)
    assert len(self._variables) == len(self._dry_threshold_mm) == len(self._min_p1) == len(self._max_p1), (
        'All arguments must have the same length.')

  @property
  def unique_name(self) -> str:
    join = lambda values: '_'.join(str(v) for v in values)
    return (f'SEEPS_{join(self._variables)}_dry_threshold_mm_{join(self._dry_threshold_mm)}_min_p1_{join(self._min_p1)}'
            f'_max_p1_{join(self._max_p1)}')

  def compute(self, predictions: Mapping[Hashable, xr.DataArray], targets: Mapping[Hashable, xr.DataArray]):
    return {v: self._one(xr.as_dataarray(predictions[v]), xr.as_dataarray(targets[v]), v, dry, lo, hi)
            for v, dry, lo, hi in zip(self._variables, self._dry_threshold_mm, self._min_p1, self._max_p1)}

  @staticmethod
  def _categories(da: xr.DataArray, wet: xr.DataArray, dry_threshold_mm: float):
    """Indicators (dry, light, heavy) as floats, NaN where the input is NaN."""
    dry_threshold = dry_threshold_mm / 1000.0   # the fields are in metres
    there = ~da.isnull()
    as_float = lambda b: b.astype(np.float64).where(there)
    return as_float(da <= dry_threshold), as_float((da > dry_threshold) & (da < wet)), as_float(da >= wet)

  def _one(self, predictions, targets, variable, dry_threshold_mm, min_p1, max_p1):
    valid_time = predictions['init_time'] + predictions['lead_time']
    wet = xr.as_dataarray(self._climatology[f'{variable}_seeps_threshold']).sel(dayofyear=valid_time.dt.dayofyear,
                                                                               hour=valid_time.dt.hour)
    p1 = xr.as_dataarray(self._climatology[f'{variable}_seeps_dry_fraction']).mean(('hour', 'dayofyear'))
    f_dry, f_light, f_heavy = self._categories(predictions, wet, dry_threshold_mm)
    o_dry, o_light, o_heavy = self._categories(targets, wet, dry_threshold_mm)
    result = 0.5 * (f_dry * o_light * (1 / (1 - p1)) + f_dry * o_heavy * (4 / (1 - p1))
                    + f_light * o_dry * (1 / p1) + f_light * o_heavy * (3 / (1 - p1))
                    + f_heavy * o_dry * (1 / p1 + 3 / (2 + p1)) + f_heavy * o_light * (3 / (2 + p1)))
    mask = (p1 >= min_p1) & (p1 <= max_p1)
    result = result.where(mask, np.nan)
    if 'mask' in predictions.coords and 'mask' in targets.coords:
      raise ValueError('Both predictions and targets have masks. This should not happen.')
    for side in (predictions, targets):
      if 'mask' in side.coords:
        mask = mask & side.coords['mask']
    mask = mask.transpose(*[d for d in result.dims if d in mask.dims])
    result = result._replace(name=variable)  # pylint: disable=protected-access
    result._coords['mask'] = (tuple(mask.dims), np.asarray(mask.values, dtype=bool))  # pylint: disable=protected-access
    return result
