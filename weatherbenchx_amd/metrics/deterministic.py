"""Deterministic statistics and metrics (counterpart of weatherbenchX/metrics/deterministic.py:91-425).

Per-point arithmetic lives in csrc/wbx_det.hip; the classes here only name the lane they need.
"""
from __future__ import annotations

from typing import Mapping, Sequence, Union

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import lazy
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base


def _sqrt(da: xr.DataArray) -> xr.DataArray:
  return np.sqrt(da)


def _clim_key(ref):
  """Identity of the climatology payload a fused group was built on (the payload object and its in-place edits)."""
  return (id(ref.source.data), ref.source.__dict__.get('_mutations', 0))


class RelativeIntensity(base.PerVariableStatistic):
  """|mean(predictions) / mean(targets) - 1| over `spatial_dims` (deterministic.py:30-88): the ratio of the spatial means, for
  non-negative fields such as precipitation.  With a `mask` coordinate on the targets the means run over mask == 1 alone (a NaN
  inside that region still makes the result NaN: the reference sums with skipna=False), a slice without any valid point gives 0,
  and the result carries `mask` = 1 where at least one point was valid.

  The spatial sums are reductions of the PASS1 family on the device (one launch per field: the unweighted sum and the count of
  the valid points per remaining index); the ratio is formed on the few numbers that are left."""

  def __init__(self, spatial_dims: Sequence[str] = ('latitude', 'longitude')):
    self._spatial_dims = spatial_dims

  def _compute_per_variable(self, predictions, targets):
    from weatherbenchx_amd import engine  # pylint: disable=g-import-not-at-top
    spatial = [str(d) for d in self._spatial_dims]
    epsilon = 1e-6  # (in numerator and denominator: no division by zero, and 0 / 0 counts as a perfect ratio)
    predictions, targets = xr.as_dataarray(predictions), xr.as_dataarray(targets)
    for name, da in (('predictions', predictions), ('targets', targets)):
      missing = [d for d in spatial if d not in da.dims]
      if missing:
        raise ValueError(f'{name} have no dimension(s) {missing} to take the spatial mean over (dims {da.dims})')
    mask = None
    if 'mask' in targets.coords:
      mc = targets.coords['mask']
      valid = mc.data == 1 if xr._is_torch(mc.data) else np.asarray(mc.values) == 1  # pylint: disable=protected-access
      mask = xr.DataArray(valid, dims=mc.dims)

    def spatial_sum(da):
      plain = xr.DataArray(da.data, dims=da.dims)
      if not xr._is_float(plain.data):  # pylint: disable=protected-access
        plain = plain.astype(np.float64)
      with engine.synchronous_results():  # (the ratio below needs the numbers now, whatever loop this runs in)
        vals, counts, out_dims = engine.reduce_statistics('det', [plain], da.dims, da.sizes, spatial, None, (), func=_hip.PASS1, mask=mask)
      coords = {d: da.coords[d] for d in out_dims if d in da.coords}
      return (xr.DataArray(np.array(vals[0], dtype=np.float64), dims=out_dims, coords=coords),
              xr.DataArray(np.array(np.broadcast_to(counts[0], np.shape(vals[0])), dtype=np.float64), dims=out_dims, coords=coords))
    psum, count = spatial_sum(predictions)
    tsum, _ = spatial_sum(targets)
    if mask is None:
      pmean, tmean = psum / count, tsum / count
    else:
      some = count.values > 0
      with np.errstate(all='ignore'):
        pmean = xr.DataArray(np.where(some, psum.values / np.where(some, count.values, 1.0), 0.0), dims=psum.dims, coords=dict(psum.coords))
        tmean = xr.DataArray(np.where(some, tsum.values / np.where(some, count.values, 1.0), 0.0), dims=tsum.dims, coords=dict(tsum.coords))
    with np.errstate(all='ignore'):
      result = abs((pmean + epsilon) / (tmean + epsilon) - 1)
    if mask is not None:
      result = result.assign_coords(mask=xr.DataArray((count.values > 0).astype(int), dims=count.dims))
    return result


class Error(base.PerVariableStatistic):
  """predictions - targets (deterministic.py:91-100)."""

  def _compute_per_variable(self, predictions, targets):
    return lazy.det_statistic('Error', predictions, targets)


class AbsoluteError(base.PerVariableStatistic):
  """|predictions - targets| (deterministic.py:103-112)."""

  def _compute_per_variable(self, predictions, targets):
    return lazy.det_statistic('AbsoluteError', predictions, targets)


class SquaredError(base.PerVariableStatistic):
  """(predictions - targets)**2 (deterministic.py:115-123)."""

  def _compute_per_variable(self, predictions, targets):
    return lazy.det_statistic('SquaredError', predictions, targets)


class PredictionPassthrough(base.PerVariableStatistic):
  """Predictions carried through with the targets' coordinates (deterministic.py:126-147)."""

  def __init__(self, copy_nans_from_targets: bool = False):
    self._copy_nans_from_targets = copy_nans_from_targets

  def _compute_per_variable(self, predictions, targets):
    out = predictions + xr.zeros_like(targets)
    return out.where(~targets.isnull()) if self._copy_nans_from_targets else out


class TargetPassthrough(base.PerVariableStatistic):
  """Targets carried through with the predictions' coordinates (deterministic.py:150-171)."""

  def __init__(self, copy_nans_from_predictions: bool = False):
    self._copy_nans_from_predictions = copy_nans_from_predictions

  def _compute_per_variable(self, predictions, targets):
    out = targets + xr.zeros_like(predictions)
    return out.where(~predictions.isnull()) if self._copy_nans_from_predictions else out


class WindVectorSquaredError(base.Statistic):
  """(u_p - u_t)**2 + (v_p - v_t)**2 per (u, v, name) triple (deterministic.py:174-219)."""

  def __init__(self, u_name: Sequence[str], v_name: Sequence[str], vector_name: Sequence[str]):
    self._u_name, self._v_name, self._vector_name = u_name, v_name, vector_name
    if not len(u_name) == len(v_name) == len(vector_name):
      raise ValueError('u_name, v_name, and vector_name must have the same length')

  @property
  def unique_name(self) -> str:
    return 'WindVectorSquaredError_' + '_'.join(self._vector_name)

  def compute(self, predictions, targets):
    out = {}
    for u, v, name in zip(self._u_name, self._v_name, self._vector_name):
      se_u = lazy.det_statistic('SquaredError', predictions[u], targets[u])
      se_v = lazy.det_statistic('SquaredError', predictions[v], targets[v])
      if se_u.dims == se_v.dims and se_u.shape == se_v.shape:
        out[name] = lazy.LinearCombination([se_u, se_v])
      else:
        out[name] = se_u + se_v
    return out


def _threshold_array(thresholds, name):
  """(values float64[K], dim name, coordinate or None) of a 1-D thresholds spec (deterministic.py:265-290)."""
  if isinstance(thresholds, (xr.Dataset, dict)):
    thresholds = thresholds[name]
  if isinstance(thresholds, xr.DataArray):
    if thresholds.ndim != 1:
      return None  # thresholds that vary with data dims: evaluated un-fused (ErrorExceedance._compute_per_variable)
    dim = thresholds.dims[0]
    coord = thresholds.coords[dim].values if dim in thresholds.coords else None
    return np.asarray(thresholds.values, np.float64), dim, coord
  values = np.asarray(list(thresholds), np.float64)
  return values, 'error_exceedance_thresholds', values


_STACKED_THRESHOLDS = '_wbx_stacked_thresholds'  # the kernel's category axis when the thresholds add 0 or >= 2 dims


class ErrorExceedance(base.PerVariableStatistic):
  """float(|p - t| > threshold) along a new thresholds dimension, NaN where the error or the threshold is NaN
  (deterministic.py:262-295).  All thresholds are lanes of one fused launch (csrc/wbx_cat.hip)."""

  def __init__(self, thresholds):
    self._thresholds = thresholds

  def _compute_per_variable(self, predictions, targets):
    spec = _threshold_array(self._thresholds, predictions.name)
    if spec is None:
      # thresholds with data dims (per level, per latitude ...): the kernel reads threshold k of a point through the field's
      # own strides (wbx_cat_exceed_field); the new dimension is the one the inputs do not have
      thresholds = self._thresholds[predictions.name] if isinstance(self._thresholds, (xr.Dataset, dict)) else self._thresholds
      thresholds = xr.as_dataarray(thresholds)
      new = [d for d in thresholds.dims if d not in predictions.dims and d not in targets.dims]
      if len(new) == 1:
        coord = thresholds.coords[new[0]].values if new[0] in thresholds.coords else None
        return lazy.cat_statistic(_hip.CAT_EXCEED, predictions, targets, new[0], coord, threshold_field=thresholds)
      # any other broadcastable field, like the reference's `abs_error > thresholds` (deterministic.py:283-295): no new dim (a
      # scalar, per-level / per-latitude thresholds) = ONE category that is squeezed out of the result; two or more new dims are
      # stacked into the kernel's category axis and split again in the result.  Same kernel, same launch count.
      shared = [d for d in thresholds.dims if d not in new]
      stacked = thresholds.transpose(*shared, *new)
      shape_new = tuple(stacked.sizes[d] for d in new)
      values = np.ascontiguousarray(np.asarray(stacked.values, np.float64)).reshape(
          tuple(stacked.sizes[d] for d in shared) + (int(np.prod(shape_new, dtype=np.int64)),))
      field = xr.DataArray(values, dims=tuple(shared) + (_STACKED_THRESHOLDS,),
                           coords={d: stacked.coords[d].values for d in shared if d in stacked.coords})
      split = (new, shape_new, {d: stacked.coords[d].values for d in new if d in stacked.coords})
      return lazy.cat_statistic(_hip.CAT_EXCEED, predictions, targets, _STACKED_THRESHOLDS, None, threshold_field=field, split=split)
    values, dim, coord = spec
    return lazy.cat_statistic(_hip.CAT_EXCEED, predictions, targets, dim, coord, thresholds=values)


class SquaredPredictionAnomaly(base.PerVariableStatisticWithClimatology):
  """(predictions - climatology)**2 (deterministic.py:222-232)."""

  def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
    return lazy.det_statistic('SquaredPredictionAnomaly', predictions, targets, aligned_climatology,
                              clim_key=_clim_key(aligned_climatology))


class SquaredTargetAnomaly(base.PerVariableStatisticWithClimatology):
  """(targets - climatology)**2 (deterministic.py:235-245)."""

  def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
    return lazy.det_statistic('SquaredTargetAnomaly', predictions, targets, aligned_climatology,
                              clim_key=_clim_key(aligned_climatology))


class AnomalyCovariance(base.PerVariableStatisticWithClimatology):
  """(predictions - climatology) * (targets - climatology) (deterministic.py:248-259)."""

  def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
    return lazy.det_statistic('AnomalyCovariance', predictions, targets, aligned_climatology,
                              clim_key=_clim_key(aligned_climatology))


# Metrics that are the plain mean of a statistic (deterministic.py:305-309).
Bias = Error
MAE = AbsoluteError
MSE = SquaredError
PredictionAverage = PredictionPassthrough
TargetAverage = TargetPassthrough


class RMSE(base.PerVariableMetric):
  """sqrt(mean SquaredError) (deterministic.py:312-324)."""

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {'SquaredError': SquaredError()}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return _sqrt(statistic_values['SquaredError'])


class WindVectorRMSE(base.Metric):
  """sqrt(mean WindVectorSquaredError) (deterministic.py:327-371)."""

  def __init__(self, u_name: Union[str, list], v_name: Union[str, list], vector_name: Union[str, list]):
    as_list = lambda x: [x] if isinstance(x, str) else x
    self._u_name, self._v_name, self._vector_name = as_list(u_name), as_list(v_name), as_list(vector_name)
    if not len(self._u_name) == len(self._v_name) == len(self._vector_name):
      raise ValueError('u_name, v_name, and vector_name must have the same length')

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {'WindVectorSquaredError': WindVectorSquaredError(self._u_name, self._v_name, self._vector_name)}

  def values_from_mean_statistics(self, statistic_values):
    return xarray_tree.map_structure(_sqrt, statistic_values['WindVectorSquaredError'])


class ACC(base.PerVariableMetric):
  """Anomaly correlation coefficient: cov / (sqrt(spa) * sqrt(sta)) (deterministic.py:374-400)."""

  def __init__(self, climatology):
    self._climatology = climatology

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    c = self._climatology
    return {'SquaredPredictionAnomaly': SquaredPredictionAnomaly(climatology=c),
            'SquaredTargetAnomaly': SquaredTargetAnomaly(climatology=c),
            'AnomalyCovariance': AnomalyCovariance(climatology=c)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return statistic_values['AnomalyCovariance'] / (
        _sqrt(statistic_values['SquaredPredictionAnomaly']) * _sqrt(statistic_values['SquaredTargetAnomaly']))


class PredictionActivity(base.PerVariableMetric):
  """Std-dev of prediction anomalies: sqrt(mean spa) (deterministic.py:403-425)."""

  def __init__(self, climatology):
    self._climatology = climatology

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    return {'SquaredPredictionAnomaly': SquaredPredictionAnomaly(climatology=self._climatology)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return _sqrt(statistic_values['SquaredPredictionAnomaly'])
