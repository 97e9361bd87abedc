"""Input transforms needed on the hot path (counterpart of weatherbenchX/metrics/wrappers.py:95-148,
967-1069): InputTransform, EnsembleMean, WrappedStatistic, RenamedStatistic, WrappedMetric -- what
`mean_rmse` of the public benchmark uses (public_benchmark/run_benchmark_evaluation.py:346-353).
The thresholding / tiling transforms are out of scope.
"""
from __future__ import annotations

import abc
from typing import Hashable, Mapping

from weatherbenchx_amd import lazy
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base


class InputTransform(abc.ABC):
  """Transformation applied to predictions and/or targets before a statistic (wrappers.py:95-113)."""

  def __init__(self, which):
    if which not in ['predictions', 'targets', 'both']:
      raise ValueError(f'Invalid value for `which`: {which}')
    self.which = which

  @property
  @abc.abstractmethod
  def unique_name_suffix(self) -> str:
    ...

  @abc.abstractmethod
  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    ...


class EnsembleMean(InputTransform):
  """Mean over the ensemble dim (wrappers.py:116-148).  The result stays lazy so that SquaredError of it is
  served by the ensemble kernel (lane 4) in the same pass as CRPS."""

  def __init__(self, which: str, ensemble_dim='number', skipna=False, skip_if_ensemble_dim_missing: bool = False):
    super().__init__(which)
    self._ensemble_dim = ensemble_dim
    self._skipna = skipna
    self._skip_if_ensemble_dim_missing = skip_if_ensemble_dim_missing

  @property
  def unique_name_suffix(self) -> str:
    # Same text as the reference's f-string with `=` specifiers (wrappers.py:143).
    return f"ensemble_mean_self._ensemble_dim={self._ensemble_dim!r}_self._skipna={self._skipna!r}"

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    da = xr.as_dataarray(da)
    if self._ensemble_dim not in da.dims and self._skip_if_ensemble_dim_missing:
      return da
    if self._ensemble_dim not in da.dims:
      raise ValueError(f'Dimension {self._ensemble_dim!r} not found in {da.dims}')
    if self._skipna:
      return da.mean(self._ensemble_dim, skipna=True)
    return lazy.LazyEnsembleMean(da, self._ensemble_dim)


class WrappedStatistic(base.Statistic):
  """A statistic evaluated on transformed inputs (wrappers.py:967-1003)."""

  def __init__(self, statistic: base.Statistic, transform: InputTransform):
    self.statistic = statistic
    self.transform = transform

  @property
  def unique_name(self) -> str:
    return f'{self.statistic.unique_name}_{self.transform.which}_{self.transform.unique_name_suffix}'

  def compute(self, predictions, targets):
    if self.transform.which in ('predictions', 'both'):
      predictions = xarray_tree.map_structure(self.transform.transform_fn, _as_tree(predictions))
    if self.transform.which in ('targets', 'both'):
      targets = xarray_tree.map_structure(self.transform.transform_fn, _as_tree(targets))
    return self.statistic.compute(predictions, targets)


def _as_tree(x):
  return x if isinstance(x, (dict, xr.Dataset)) else dict(x)


class RenamedStatistic(base.Statistic):
  """A statistic under another unique name (wrappers.py:1006-1022)."""

  def __init__(self, statistic: base.Statistic, unique_name: str):
    self._statistic = statistic
    self._unique_name = unique_name

  @property
  def unique_name(self) -> str:
    return self._unique_name

  def compute(self, predictions, targets):
    return self._statistic.compute(predictions, targets)


class WrappedMetric(base.Metric):
  """All statistics of a metric behind a list of transforms, applied in list order (wrappers.py:1025-1069)."""

  def __init__(self, metric: base.Metric, transforms: list, unique_name_suffix: str | None = None):
    self.metric = metric
    self.transforms = transforms
    self.unique_name_suffix = unique_name_suffix

  @property
  def statistics(self) -> Mapping[Hashable, base.Statistic]:
    out = {}
    for name, stat in self.metric.statistics.items():
      original = stat.unique_name
      for transform in reversed(self.transforms):  # outermost wrapper runs first
        stat = WrappedStatistic(stat, transform)
      if self.unique_name_suffix is not None:
        stat = RenamedStatistic(stat, f'{original}_{self.unique_name_suffix}')
      out[name] = stat
    return out

  def values_from_mean_statistics(self, statistic_values):
    return self.metric.values_from_mean_statistics(statistic_values)
