"""Input transforms needed on the hot path (counterpart of weatherbenchX/metrics/wrappers.py:95-148,
967-1069): InputTransform, EnsembleMean, WrappedStatistic, RenamedStatistic, WrappedMetric -- what
`mean_rmse` of the public benchmark uses (public_benchmark/run_benchmark_evaluation.py:346-353).
Also the small input transforms that are plain labeled-array operations -- Inline, ReLU, Rename, Select, ContinuousToBinary,
EnsembleQuantiles, ShiftAlongNewDim, WeibullEnsembleToProbabilistic (wrappers.py:50-89, 151-267, 550-742, 745-808) -- and SubselectVariables (wrappers.py:1072-1120): they run on the arrays as they
are (host or HBM) and hand the statistics ordinary inputs; likewise the CDF / bin indicators with thresholds that follow a
chunk's time labels (wrappers.py:270-547), StackToNewDimension and Tile (wrappers.py:811-964).
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


# ---- CDF / bin indicators, stacking, tiles --------------------------------------------------------------------------------------------
def _take_along(da: xr.DataArray, dim: str, labels: np.ndarray, label_dims, label_coords) -> xr.DataArray:
  """da.sel({dim: labels}) for an indexer of any rank: `dim` is replaced by `label_dims`; the labels stay as a coordinate `dim`."""
  da = xr.as_dataarray(da)
  have = np.asarray(da.coords[dim].values)
  labels = np.asarray(labels)
  if have.dtype.kind in 'mM' or labels.dtype.kind in 'mM':
    have, want = have.astype('datetime64[ns]' if have.dtype.kind == 'M' else 'timedelta64[ns]').astype(np.int64), \
        labels.astype('datetime64[ns]' if labels.dtype.kind == 'M' else 'timedelta64[ns]').astype(np.int64)
  else:
    want = labels
  order = np.argsort(have, kind='stable')
  pos = np.searchsorted(have[order], want.reshape(-1))
  pos = np.clip(pos, 0, len(have) - 1)
  idx = order[pos]
  if not np.array_equal(have[idx], want.reshape(-1)):
    missing = want.reshape(-1)[have[idx] != want.reshape(-1)][:3]
    raise KeyError(f'not all values found in index {dim!r}: {missing}')
  axis = da.dims.index(dim)
  data = da.data
  if xr._is_torch(data):  # pylint: disable=protected-access
    import torch  # pylint: disable=g-import-not-at-top
    taken = torch.index_select(data, axis, torch.as_tensor(idx, device=data.device))
  else:
    taken = np.take(np.asarray(data), idx, axis=axis)
  shape = tuple(da.shape[:axis]) + tuple(labels.shape) + tuple(da.shape[axis + 1:])
  taken = taken.reshape(shape)
  dims = tuple(da.dims[:axis]) + tuple(label_dims) + tuple(da.dims[axis + 1:])
  coords = {k: v for k, v in da.coords.items() if dim not in v.dims}
  coords.update(label_coords)
  coords[dim] = (tuple(label_dims), np.asarray(da.coords[dim].values)[idx].reshape(labels.shape))
  return xr.DataArray(taken, dims=dims, coords=coords, name=da.name, attrs=da.attrs)


def _dayofyear(t: np.ndarray) -> np.ndarray:
  t = np.asarray(t).astype('datetime64[ns]')
  return (t.astype('datetime64[D]') - t.astype('datetime64[Y]').astype('datetime64[D]')).astype(np.int64) + 1


def select_bin_thresholds_by_time_from_chunk(bin_thresholds: xr.DataArray, chunk: xr.DataArray) -> xr.DataArray:
  """The thresholds that belong to a chunk's time labels (wrappers.py:270-346).  The chunk has (init_time, lead_time), or
  valid_time / time, or no time at all (thresholds unchanged); the thresholds are indexed by valid_time / time, by (init_time,
  lead_time), by dayofyear (+ lead_time), or by nothing."""
  bt, chunk = xr.as_dataarray(bin_thresholds), xr.as_dataarray(chunk)
  cc = chunk.coords

  def coord(name):
    c = cc[name]
    return np.asarray(c.values), tuple(c.dims), {k: v for k, v in cc.items() if set(v.dims) <= set(c.dims) and k in c.dims}
  if 'init_time' in cc and 'lead_time' in cc:
    it, idims, icoords = coord('init_time')
    lt, ldims, lcoords = coord('lead_time')
    if idims == ldims:  # sparse: both on one dim
      valid, vdims = it + lt, idims
    else:
      valid, vdims = it[:, None] + lt[None, :], idims + ldims
    vcoords = {**icoords, **lcoords}
    for name in ('valid_time', 'time'):
      if name in bt.dims:
        return _take_along(bt, name, valid, vdims, vcoords)
    if {'init_time', 'lead_time'} <= set(bt.dims):
      return _take_along(_take_along(bt, 'init_time', it, idims, icoords), 'lead_time', lt, ldims, lcoords)
    if 'dayofyear' in bt.dims:
      if 'lead_time' in bt.dims:
        return _take_along(_take_along(bt, 'dayofyear', _dayofyear(it), idims, icoords), 'lead_time', lt, ldims, lcoords)
      return _take_along(bt, 'dayofyear', _dayofyear(valid), vdims, vcoords)
    return bt
  for name in ('valid_time', 'time'):
    if name in cc:
      t, tdims, tcoords = coord(name)
      if tdims != (name,):
        tcoords = {name: (tdims, t), **{k: v for k, v in cc.items() if k in tdims}}
      for tname in ('valid_time', 'time'):
        if tname in bt.dims:
          return _take_along(bt, tname, t, tdims, tcoords)
      if 'dayofyear' in bt.dims:
        return _take_along(bt, 'dayofyear', _dayofyear(t), tdims, tcoords)
      return bt
  return bt


def compute_cdf(threshold_values, da: xr.DataArray, threshold_dim: str, enforce_monotonicity: bool, right_inclusive: bool = True) -> xr.DataArray:
  """[x <= threshold] (or <) for every threshold as float, NaN where x or the threshold is NaN (wrappers.py:349-390)."""
  da = xr.as_dataarray(da)
  if isinstance(threshold_values, (xr.DataArray, xr.Dataset)):
    thresholds = threshold_values[da.name] if isinstance(threshold_values, xr.Dataset) else threshold_values
    thresholds = select_bin_thresholds_by_time_from_chunk(thresholds, da)
  elif isinstance(threshold_values, Iterable):
    values = np.array(list(threshold_values))
    thresholds = xr.DataArray(values, dims=[threshold_dim], coords={threshold_dim: values})
  else:
    raise ValueError('Bin values must be an Iterable, xr.DataArray, or xr.Dataset.')
  if enforce_monotonicity:
    tv = np.asarray(thresholds.values)
    if not np.all(np.diff(tv, axis=thresholds.dims.index(threshold_dim)) > 0):
      raise ValueError('Bin values must be monotonically increasing. To turn off this check, set `enforce_monotonicity=False`.')
  cdf = ((da <= thresholds) if right_inclusive else (da < thresholds)).astype(np.float64)
  return cdf.where(~da.isnull()).where(~thresholds.isnull())


class ContinuousToCDF(InputTransform):
  """A continuous input as the indicators [x <= t] (or [x < t]) along `threshold_dim` (wrappers.py:480-547)."""

  def __init__(self, which: str, threshold_values, threshold_dim: str, unique_name_suffix: str | None = None,
               enforce_monotonicity: bool = True, right_inclusive: bool = True):
    super().__init__(which)
    self._threshold_values = threshold_values
    self._threshold_dim = threshold_dim
    if isinstance(threshold_values, (xr.DataArray, xr.Dataset)) and unique_name_suffix is None:
      raise ValueError('unique_name_suffix must be provided if threshold_values is an xarray.DataArray or xarray.Dataset.')
    self._unique_name_suffix = unique_name_suffix
    self._enforce_monotonicity = enforce_monotonicity
    self._right_inclusive = right_inclusive

  @property
  def unique_name_suffix(self) -> str:
    suffix = self._unique_name_suffix if self._unique_name_suffix is not None else ','.join(str(t) for t in self._threshold_values)
    return f'ContinuousToCDF_{self._threshold_dim}_{suffix}_right_inclusive_{self._right_inclusive}'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    return compute_cdf(self._threshold_values, da, self._threshold_dim, self._enforce_monotonicity, self._right_inclusive)


class ContinuousToBins(InputTransform):
  """A continuous input as indicators of the right-inclusive bins (t[i-1], t[i]] along `bin_dim` -- len(bin_values) - 1 of them,
  +-inf at the ends for open bins -- labelled '{left:.2f} < p <= {right:.2f}' with `{bin_dim}_left` / `_right` coordinates
  (wrappers.py:393-477)."""

  def __init__(self, which: str, bin_values, bin_dim: str, unique_name_suffix: str | None = None, enforce_monotonicity: bool = True):
    super().__init__(which)
    self._bin_values = bin_values
    self._bin_dim = bin_dim
    if isinstance(bin_values, (xr.DataArray, xr.Dataset)) and unique_name_suffix is None:
      raise ValueError('unique_name_suffix must be provided if bin_values is an xarray.DataArray or xarray.Dataset.')
    self._unique_name_suffix = unique_name_suffix
    self._enforce_monotonicity = enforce_monotonicity

  @property
  def unique_name_suffix(self) -> str:
    suffix = self._unique_name_suffix if self._unique_name_suffix is not None else ','.join(str(t) for t in self._bin_values)
    return f'ContinuousToBins_{self._bin_dim}_{suffix}'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    cdf = compute_cdf(self._bin_values, da, self._bin_dim, self._enforce_monotonicity)
    edges = np.asarray(cdf.coords[self._bin_dim].values)
    left, right = edges[:-1], edges[1:]
    n = cdf.sizes[self._bin_dim]
    upper = cdf.isel({self._bin_dim: slice(1, n)})
    lower = cdf.isel({self._bin_dim: slice(0, n - 1)})
    # (the difference of two slices whose labels differ: put the same labels on both first)
    names = np.array([f'{a:.2f} < p <= {b:.2f}' for a, b in zip(left, right)])
    upper, lower = upper.assign_coords({self._bin_dim: names}), lower.assign_coords({self._bin_dim: names})
    out = upper - lower
    return out.assign_coords({f'{self._bin_dim}_left': ((self._bin_dim,), left), f'{self._bin_dim}_right': ((self._bin_dim,), right)})


class StackToNewDimension(InputTransform):
  """Several dims flattened (in the order given, the last one fastest) into one dim labelled 0 .. n - 1, whose name may be one of
  theirs (wrappers.py:811-848)."""

  def __init__(self, which: str, dims_to_stack: Sequence[Hashable], new_dim_name: Hashable):
    super().__init__(which)
    self._dims_to_stack = dims_to_stack
    self._new_dim_name = new_dim_name

  @property
  def unique_name_suffix(self) -> str:
    return f'stack_{self._dims_to_stack}_to_{self._new_dim_name}'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    da = xr.as_dataarray(da)
    stack = list(self._dims_to_stack)
    keep = [d for d in da.dims if d not in stack]
    moved = da.transpose(*keep, *stack)
    n = int(np.prod([da.sizes[d] for d in stack], dtype=np.int64))
    data = moved.data.reshape(tuple(da.sizes[d] for d in keep) + (n,))
    coords = {k: v for k, v in da.coords.items() if not set(v.dims) & set(stack)}
    coords[self._new_dim_name] = np.arange(n)
    return xr.DataArray(data, dims=tuple(keep) + (self._new_dim_name,), coords=coords, name=da.name, attrs=da.attrs)


def construct_tiles(da: xr.DataArray, window_size: int = 3, window_dim: str = 'window', wrap_longitude: bool = False) -> xr.DataArray:
  """At every (latitude, longitude) pixel the window_size x window_size patch around it, along a new leading `window_dim`
  (latitude offset slowest); pixels whose patch would cross the latitude edges -- and, unless `wrap_longitude`, the longitude edges
  -- are dropped (wrappers.py:851-909)."""
  da = xr.as_dataarray(da)
  half = window_size // 2
  lo, hi = half, window_size - 1 - half
  ilat, ilon = da.dims.index('latitude'), da.dims.index('longitude')
  data = da.data
  roll = (lambda x, s, ax: x.roll(s, ax)) if xr._is_torch(data) else (lambda x, s, ax: np.roll(x, s, axis=ax))  # pylint: disable=protected-access
  layers = []
  for i in range(window_size):
    for j in range(window_size):
      layers.append(roll(roll(data, i - half, ilat), j - half, ilon))
  if xr._is_torch(data):  # pylint: disable=protected-access
    import torch  # pylint: disable=g-import-not-at-top
    stacked = torch.stack(layers)
  else:
    stacked = np.stack(layers)
  out = xr.DataArray(stacked, dims=(window_dim,) + tuple(da.dims), coords=dict(da.coords), name=da.name, attrs=da.attrs)
  out = out.isel(latitude=slice(lo, da.sizes['latitude'] - hi))
  if not wrap_longitude:
    out = out.isel(longitude=slice(lo, da.sizes['longitude'] - hi))
  return out


class Tile(InputTransform):
  """construct_tiles as an input transform (wrappers.py:912-964)."""

  def __init__(self, which: str, window_size: int = 3, window_dim: str = 'window', wrap_longitude: bool = False):
    super().__init__(which)
    self._window_size = window_size
    self._window_dim = window_dim
    self._wrap_longitude = wrap_longitude

  @property
  def unique_name_suffix(self) -> str:
    return f'tiled_window_size_{self._window_size}_wrap_{self._wrap_longitude}_dim_{self._window_dim}'

  def transform_fn(self, da: xr.DataArray) -> xr.DataArray:
    return construct_tiles(da, window_size=self._window_size, window_dim=self._window_dim, wrap_longitude=self._wrap_longitude)
