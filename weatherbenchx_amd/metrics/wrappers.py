"""Input transforms needed on the hot path (counterpart of weatherbenchX/metrics/wrappers.py:95-148,
967-1069): InputTransform, EnsembleMean, WrappedStatistic, RenamedStatistic, WrappedMetric -- what
`mean_rmse` of the public benchmark uses (public_benchmark/run_benchmark_evaluation.py:346-353).
Also the small input transforms that are plain labeled-array operations -- Inline, ReLU, Rename, Select, ContinuousToBinary,
EnsembleQuantiles, ShiftAlongNewDim, WeibullEnsembleToProbabilistic (wrappers.py:50-89, 151-267, 550-742, 745-808) -- and SubselectVariables (wrappers.py:1072-1120): they run on the arrays as they
are (host or HBM) and hand the statistics ordinary inputs.  The binning / CDF / tiling / stacking transforms are out of scope.
"""
from __future__ import annotations

import abc
from typing import Any, Callable, Hashable, Iterable, Mapping, Sequence

import numpy as np

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


# ---- small input transforms: plain labeled-array operations ---------------------------------------------------------------------------
def binarize_thresholds(x: xr.DataArray, thresholds, threshold_dim: str) -> xr.DataArray:
  """x > threshold for every threshold along a new (or the thresholds' own) `threshold_dim`; NaN stays NaN, so the result is
  float32 (wrappers.py:50-89).  `thresholds`: values, a DataArray with `threshold_dim`, or a Dataset with one such array per
  variable name."""
  x = xr.as_dataarray(x)
  if isinstance(thresholds, xr.Dataset):
    assert threshold_dim in thresholds.dims, f'threshold_dim ({threshold_dim}) not found in thresholds ({thresholds.dims})'
    assert x.name in thresholds.data_vars, f'Input DataArray name ({x.name}) not found in thresholds ({list(thresholds.data_vars)})'
    threshold = thresholds[x.name]
  elif isinstance(thresholds, xr.DataArray):
    assert threshold_dim in thresholds.dims, f'threshold_dim ({threshold_dim}) not found in thresholds ({thresholds.dims})'
    threshold = thresholds
  else:
    values = np.asarray(list(thresholds))
    threshold = xr.DataArray(values, dims=[threshold_dim], coords={threshold_dim: values})
  out = (x > threshold).where(~x.isnull()).astype(np.float32)
  return out.rename(x.name) if x.name is not None else out


class ContinuousToBinary(InputTransform):
  """A continuous input as exceedance indicators along `threshold_dim` (wrappers.py:214-267)."""

  def __init__(self, which: str, threshold_value, threshold_dim: str, unique_name_suffix: str | None = None):
    super().__init__(which)
    labeled = isinstance(threshold_value, (xr.DataArray, xr.Dataset))
    self._threshold_value = threshold_value if (labeled or isinstance(threshold_value, Iterable)) else [threshold_value]
    self._threshold_dim = threshold_dim
    if labeled and unique_name_suffix is None:
      raise ValueError('unique_name_suffix must be provided if threshold_value is an xarray.DataArray or xarray.Dataset.')
    self._unique_name_suffix = unique_name_suffix

  @property
  def unique_name_suffix(self) -> str:
    suffix = self._unique_name_suffix
    if suffix is None:
      suffix = ','.join(str(t) for t in self._threshold_value)
    return f'{self._threshold_dim}={suffix}'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    return binarize_thresholds(da, self._threshold_value, self._threshold_dim)


class EnsembleQuantiles(InputTransform):
  """Quantiles over the ensemble dim along a new leading `quantile_dim` (wrappers.py:151-211; numpy's linear interpolation, as
  xarray's default).  `skipna` takes the quantiles of the members that are there (a point without any stays NaN)."""

  def __init__(self, which: str, quantiles, quantile_dim: str = 'quantile', ensemble_dim: str = 'number', skipna: bool = False,
               skip_if_ensemble_dim_missing: bool = False):
    super().__init__(which)
    self._quantiles = quantiles if isinstance(quantiles, Iterable) else [quantiles]
    self._quantile_dim = quantile_dim
    self._ensemble_dim = ensemble_dim
    self._skipna = skipna
    self._skip_if_ensemble_dim_missing = skip_if_ensemble_dim_missing

  @property
  def unique_name_suffix(self) -> str:
    quantiles_str = ','.join(str(q) for q in self._quantiles)
    return (f'ensemble_quantiles_self._ensemble_dim={self._ensemble_dim!r}_self._quantile_dim={self._quantile_dim!r}_'
            f'self._skipna={self._skipna!r}_{quantiles_str}')

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    da = xr.as_dataarray(da)
    if self._ensemble_dim not in da.dims and self._skip_if_ensemble_dim_missing:
      return da
    if 'quantile' in da.dims:
      raise ValueError('Input DataArray already has a `quantile` dimension. Please rename it before applying the EnsembleQuantiles '
                       'wrapper.')
    if self._ensemble_dim not in da.dims:
      raise ValueError(f'Dimension {self._ensemble_dim!r} not found in {da.dims}')
    q = np.asarray(list(self._quantiles), dtype=np.float64)
    axis = da.dims.index(self._ensemble_dim)
    if xr._is_torch(da.data):  # pylint: disable=protected-access
      import torch  # pylint: disable=g-import-not-at-top  (a payload in HBM stays there: torch's own quantile kernels)
      data = da.data if da.data.is_floating_point() else da.data.double()
      qt = torch.as_tensor(q, dtype=data.dtype, device=data.device)
      out = (torch.nanquantile if self._skipna else torch.quantile)(data, qt, dim=axis)
    else:
      values = np.asarray(da.values)
      with np.errstate(all='ignore'):
        import warnings  # pylint: disable=g-import-not-at-top
        with warnings.catch_warnings():
          warnings.simplefilter('ignore', RuntimeWarning)  # (all-NaN slices)
          out = (np.nanquantile if self._skipna else np.quantile)(values, q, axis=axis)
    rest = tuple(d for d in da.dims if d != self._ensemble_dim)
    coords = {k: v for k, v in da.coords.items() if self._ensemble_dim not in v.dims}
    coords[self._quantile_dim] = q
    return xr.DataArray(out, dims=(self._quantile_dim,) + rest, coords=coords, name=da.name, attrs=da.attrs)


class WeibullEnsembleToProbabilistic(InputTransform):
  """Binarised members -> probability by Weibull's plotting position, sum / (M + 1) (wrappers.py:550-584; Makkonen 2006)."""

  def __init__(self, which, ensemble_dim='number', skipna=False):
    assert which == 'predictions', 'Only predictions can be converted to probabilities'
    super().__init__(which)
    self._ensemble_dim = ensemble_dim
    self._skipna = skipna

  @property
  def unique_name_suffix(self) -> str:
    return 'ensemble_to_probabilistic_by_weibull_plotting_position'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    da = xr.as_dataarray(da)
    members = da.sizes[self._ensemble_dim]
    return da.sum(self._ensemble_dim, skipna=self._skipna) / (members + 1)


class ShiftAlongNewDim(InputTransform):
  """x + shift for every shift along a new dim: constants, or per-variable fields from a Dataset that already carries `shift_dim`
  (wrappers.py:648-742)."""

  def __init__(self, which: str, shift_value, shift_dim: str, unique_name_suffix: str):
    super().__init__(which)
    self._shift_value = shift_value if isinstance(shift_value, (Iterable, xr.Dataset)) else [shift_value]
    self._shift_dim = shift_dim
    self._unique_name_suffix = unique_name_suffix

  @property
  def unique_name_suffix(self) -> str:
    return self._unique_name_suffix

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    da = xr.as_dataarray(da)
    if isinstance(self._shift_value, xr.Dataset):
      shifts = self._shift_value[da.name]
      if self._shift_dim not in shifts.dims:
        raise RuntimeError(f'Expected to find self._shift_dim={self._shift_dim!r} in shifts.dims={shifts.dims!r} but did not. This is '
                           'probably an error.')
    else:
      values = np.asarray(list(self._shift_value))
      shifts = xr.DataArray(values, dims=[self._shift_dim], coords={self._shift_dim: values})
    return da + shifts


class Inline(InputTransform):
  """Any function of a DataArray, under a name of the caller's choosing (wrappers.py:587-622)."""

  def __init__(self, which: str, transform_fn: Callable[[xr.DataArray], xr.DataArray], unique_name_suffix: str):
    super().__init__(which)
    self._transform_fn = transform_fn
    self._unique_name_suffix = unique_name_suffix

  @property
  def unique_name_suffix(self) -> str:
    return f'{self._unique_name_suffix}'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    return self._transform_fn(da)


class ReLU(InputTransform):
  """max(x, 0) with NaN kept (wrappers.py:625-645)."""

  @property
  def unique_name_suffix(self) -> str:
    return 'relu'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    da = xr.as_dataarray(da)
    return da.where(da > 0, 0).where(~da.isnull())


class Rename(InputTransform):
  """Renames the variable, coordinates and dimensions (wrappers.py:745-768)."""

  def __init__(self, which: str, renames: Mapping[Hashable, Hashable]):
    super().__init__(which)
    self._renames = renames

  @property
  def unique_name_suffix(self) -> str:
    return f'rename_{self._renames}'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    return xr.as_dataarray(da).rename(self._renames)


class Select(InputTransform):
  """`.sel` and / or `.isel` on the inputs (wrappers.py:771-808)."""

  def __init__(self, which: str, sel: Mapping[Hashable, Any] | None = None, isel: Mapping[Hashable, Any] | None = None,
               sel_kwargs: Mapping[Hashable, Any] | None = None, isel_kwargs: Mapping[Hashable, Any] | None = None):
    super().__init__(which)
    self._isel = isel
    self._sel = sel
    self._isel_kwargs = isel_kwargs or {}
    self._sel_kwargs = sel_kwargs or {}

  @property
  def unique_name_suffix(self) -> str:
    # (the reference's f-string with `=` specifiers, spelled out)
    return (f'select_self._isel={self._isel!r}_self._isel_kwargs={self._isel_kwargs!r}_self._sel={self._sel!r}'
            f'_self._sel_kwargs={self._sel_kwargs!r}')

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    da = xr.as_dataarray(da)
    if self._sel is not None:
      da = da.sel(self._sel, **self._sel_kwargs)
    if self._isel is not None:
      da = da.isel(self._isel, **self._isel_kwargs)
    return da


class SubselectVariablesForStatistic(base.Statistic):
  """A statistic for some of the variables only (wrappers.py:1072-1099)."""

  def __init__(self, statistic: base.Statistic, variables: Sequence[str]):
    self.statistic = statistic
    self.variables = variables

  @property
  def unique_name(self) -> str:
    return f'{self.statistic.unique_name}_{"_".join(self.variables)}'

  def compute(self, predictions, targets):
    return self.statistic.compute({k: v for k, v in predictions.items() if k in self.variables},
                                  {k: v for k, v in targets.items() if k in self.variables})


class SubselectVariables(base.Metric):
  """A metric for some of the variables only (wrappers.py:1102-1127)."""

  def __init__(self, metric: base.Metric, variables: Sequence[str]):
    self.metric = metric
    self.variables = variables

  @property
  def statistics(self) -> Mapping[Hashable, base.Statistic]:
    return {name: SubselectVariablesForStatistic(stat, self.variables) for name, stat in self.metric.statistics.items()}

  def values_from_mean_statistics(self, statistic_values):
    return self.metric.values_from_mean_statistics(statistic_values)


# (deprecated in the reference: PerVariableStatistic / PerVariableMetric intersect the variables themselves, wrappers.py:1130-1134)
IntersectPredictionAndTargetVariablesForStatistic = lambda statistic: statistic  # pylint: disable=invalid-name
IntersectPredictionAndTargetVariables = lambda metric: metric  # pylint: disable=invalid-name
