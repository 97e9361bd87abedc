"""Fractions skill score and its neighbourhood averaging (counterpart of weatherbenchX/metrics/spatial.py:24-339).

Outside the path SURVEY section 8 names and without a kernel of its own: the neighbourhood means are sliding-window sums on whatever
holds the payload (NumPy on the host, torch for a chunk in HBM), the three per-point statistics are squares of them, and their
weighted, binned, masked means are the Aggregator's reduction like every other statistic.

The reference convolves with `scipy.ndimage.convolve1d` (two passes of n taps per point, field by field through
`xr.apply_ufunc(vectorize=True)`).  Here a window sum is a difference of two running sums -- independent of n -- over all fields of
the array at once, with NaNs counted separately so that a window holding a NaN is NaN and nothing else is.
"""
from __future__ import annotations

import dataclasses
from typing import Iterable, Mapping, Union

import numpy as np

from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base

_SPATIAL = ('latitude', 'longitude')


def _torch_like(x) -> bool:
  return xr._is_torch(x)  # pylint: disable=protected-access


def _window_mean(x, n: int, axis: int):
  """Mean over the n cyclic neighbours along `axis` (float64); NaN where the window holds a NaN."""
  size = x.shape[axis]
  half = (n - 1) // 2
  index = np.arange(-half, size + half) % size                       # the cyclically extended axis
  if _torch_like(x):
    import torch  # pylint: disable=g-import-not-at-top
    ext = x.to(torch.float64).index_select(axis, torch.as_tensor(index, device=x.device))
    missing = torch.isnan(ext)
    zero = torch.zeros_like(ext.narrow(axis, 0, 1))
    run = lambda a: torch.cat([zero, a.cumsum(dim=axis)], dim=axis)
    total, holes = run(torch.where(missing, torch.zeros_like(ext), ext)), run(missing.to(torch.float64))
    window = lambda c: c.narrow(axis, n, size) - c.narrow(axis, 0, size)
    mean = window(total) / n
    return torch.where(window(holes) > 0, torch.full_like(mean, float('nan')), mean)
  ext = np.take(np.asarray(x, dtype=np.float64), index, axis=axis)
  missing = np.isnan(ext)
  zero = np.zeros_like(np.take(ext, [0], axis=axis))
  run = lambda a: np.concatenate([zero, np.cumsum(a, axis=axis)], axis=axis)
  total, holes = run(np.where(missing, 0.0, ext)), run(missing.astype(np.float64))
  window = lambda c: np.take(c, np.arange(n, n + size), axis=axis) - np.take(c, np.arange(size), axis=axis)
  return np.where(window(holes) > 0, np.nan, window(total) / n)


def convolve2d_wrap_longitude(x, neighborhood_size: int, wrap_longitude: bool = False):
  """n x n neighbourhood mean over the last two axes (latitude, longitude) as float32.  Both axes are treated cyclically and the
  `half = (n - 1) / 2` outermost rows are then set to 0 -- as are the outermost columns unless `wrap_longitude` -- which the FSS
  reads as "nothing there"; n = 1 returns the input itself.  spatial.py:24-56."""
  if neighborhood_size == 1:
    return x
  if neighborhood_size % 2 != 1:
    raise ValueError('neighborhood_size must be odd.')
  half = (neighborhood_size - 1) // 2
  out = _window_mean(_window_mean(x, neighborhood_size, -2), neighborhood_size, -1)
  if _torch_like(out):
    import torch  # pylint: disable=g-import-not-at-top
    out = out.to(torch.float32)
  else:
    out = out.astype(np.float32)
  out[..., :half, :] = 0
  out[..., out.shape[-2] - half:, :] = 0
  if not wrap_longitude:
    out[..., :, :half] = 0
    out[..., :, out.shape[-1] - half:] = 0
  return out


def neighborhood_averaging_for_single_size(da: xr.DataArray, neighborhood_size: int, wrap_longitude: bool = False) -> xr.DataArray:
  """`convolve2d_wrap_longitude` of every (latitude, longitude) field of `da`; the spatial dims end up last.  A `mask` coordinate
  is averaged the same way and stays True only where the whole neighbourhood was (spatial.py:59-81)."""
  da = xr.as_dataarray(da)
  rest = tuple(d for d in da.dims if d not in _SPATIAL)
  moved = da.transpose(*rest, *_SPATIAL)
  payload = moved.data
  if not _torch_like(payload):
    payload = np.array(payload, copy=True)
  elif neighborhood_size == 1:
    payload = payload.clone()
  out = moved._replace(data=convolve2d_wrap_longitude(payload, neighborhood_size, wrap_longitude))  # pylint: disable=protected-access
  if 'mask' in da.coords:
    mask = da.coords['mask']
    mask = xr.DataArray(np.asarray(mask.values), dims=mask.dims)
    averaged = neighborhood_averaging_for_single_size(mask, neighborhood_size, wrap_longitude)
    out._coords['mask'] = (tuple(averaged.dims), np.isclose(np.asarray(averaged.values, dtype=np.float64), 1.0))  # pylint: disable=protected-access
  return out


def neighborhood_averaging(da: xr.DataArray, neighborhood_size: Union[int, Iterable[int]], wrap_longitude: bool = False):
  """One size, or several along a new leading `neighborhood_size` dim (spatial.py:84-102)."""
  if isinstance(neighborhood_size, Iterable):
    sizes = list(neighborhood_size)
    parts = [neighborhood_averaging_for_single_size(da, n, wrap_longitude) for n in sizes]
    masks = [p._coords.pop('mask', None) for p in parts]  # pylint: disable=protected-access
    parts = [p.expand_dims(neighborhood_size=[n]) for p, n in zip(parts, sizes)]
    out = xr.concat(parts, dim='neighborhood_size')
    if masks and masks[0] is not None:
      out._coords['mask'] = (('neighborhood_size',) + masks[0][0], np.stack([np.asarray(m[1]) for m in masks]))  # pylint: disable=protected-access
    return out
  return neighborhood_averaging_for_single_size(da, neighborhood_size, wrap_longitude)


def get_fss_mask(predictions: xr.DataArray, targets: xr.DataArray, neighborhood_size: Union[int, Iterable[int]],
                 wrap_longitude: bool = False, combine_mask: bool = False):
  """Where an FSS term is valid: the `mask` coordinate (of the targets, else of the predictions; their conjunction with
  `combine_mask`) averaged over the neighbourhood and still 1, i.e. the whole neighbourhood valid and not a zeroed boundary.
  None when neither input carries a mask.  spatial.py:105-157."""
  predictions, targets = xr.as_dataarray(predictions), xr.as_dataarray(targets)
  pm = predictions.coords['mask'] if 'mask' in predictions.coords else None
  tm = targets.coords['mask'] if 'mask' in targets.coords else None
  if pm is None and tm is None:
    return None
  if combine_mask and pm is not None and tm is not None:
    mask = pm & tm
  else:
    mask = tm if tm is not None else pm
  mask = xr.DataArray(np.asarray(mask.values, dtype=bool), dims=mask.dims)
  averaged = neighborhood_averaging(mask, neighborhood_size, wrap_longitude)
  return abs(averaged.astype(np.float64) - 1.0) < 1e-5


def get_suffix(neighborhood_size: Union[int, Iterable[int]], wrap_longitude: bool = False) -> str:
  suffix = ','.join(str(t) for t in neighborhood_size) if isinstance(neighborhood_size, Iterable) else str(neighborhood_size)
  return suffix + ('_wrap_longitude' if wrap_longitude else '')


def _without_mask(da: xr.DataArray) -> xr.DataArray:
  return da.drop_vars(['mask']) if 'mask' in da.coords else da


@dataclasses.dataclass
class _FractionStatistic(base.PerVariableStatistic):
  neighborhood_size_in_pixels: Union[int, Iterable[int]]
  wrap_longitude: bool = False
  combine_mask: bool = False

  @property
  def unique_name(self) -> str:
    return f'{type(self).__name__}_{get_suffix(self.neighborhood_size_in_pixels, self.wrap_longitude)}'

  def _fractions(self, da):
    return _without_mask(neighborhood_averaging(da, self.neighborhood_size_in_pixels, self.wrap_longitude))

  def _term(self, predictions, targets):
    raise NotImplementedError

  def _compute_per_variable(self, predictions, targets):
    predictions, targets = xr.as_dataarray(predictions), xr.as_dataarray(targets)
    mask = get_fss_mask(predictions, targets, self.neighborhood_size_in_pixels, self.wrap_longitude, self.combine_mask)
    result = self._term(predictions, targets)
    if mask is not None:
      mask = mask.transpose(*[d for d in result.dims if d in mask.dims])
      result._coords['mask'] = (tuple(mask.dims), np.asarray(mask.values, dtype=bool))  # pylint: disable=protected-access
    return result


@dataclasses.dataclass
class SquaredFractionsError(_FractionStatistic):
  """(neighbourhood fraction of the predictions - of the targets)**2: the numerator of the FSS.  spatial.py:172-207."""

  def _term(self, predictions, targets):
    diff = self._fractions(predictions) - self._fractions(targets)
    return diff * diff


@dataclasses.dataclass
class SquaredPredictionFraction(_FractionStatistic):
  """Squared neighbourhood fraction of the predictions.  spatial.py:210-242."""

  def _term(self, predictions, targets):
    del targets
    f = self._fractions(predictions)
    return f * f


@dataclasses.dataclass
class SquaredTargetFraction(_FractionStatistic):
  """Squared neighbourhood fraction of the targets.  spatial.py:245-277."""

  def _term(self, predictions, targets):
    del predictions
    f = self._fractions(targets)
    return f * f


@dataclasses.dataclass
class FSS(base.PerVariableMetric):
  """Fractions skill score (Roberts & Lean 2008) of binary fields: 1 - mean (Pf - Tf)**2 / (mean Pf**2 + mean Tf**2) with Pf, Tf the
  fractions of events in an n x n pixel neighbourhood (n odd; a list adds a `neighborhood_size` dim).  NaN when there is no event at
  all.  spatial.py:280-339."""
  neighborhood_size_in_pixels: Union[int, Iterable[int]]
  wrap_longitude: bool = False
  combine_mask: bool = False

  @property
  def statistics(self) -> Mapping[str, base.Statistic]:
    args = (self.neighborhood_size_in_pixels, self.wrap_longitude, self.combine_mask)
    return {c.__name__: c(*args) for c in (SquaredFractionsError, SquaredPredictionFraction, SquaredTargetFraction)}

  def _values_from_mean_statistics_per_variable(self, statistic_values):
    return 1 - statistic_values['SquaredFractionsError'] / (
        statistic_values['SquaredPredictionFraction'] + statistic_values['SquaredTargetFraction'])
