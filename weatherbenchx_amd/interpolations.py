"""Chunk pre-processing between loader and statistics (counterpart of weatherbenchX/interpolations.py:27-488): regridding to
fixed or reference coordinates (gridded or station-like), cropping, subsampling, neighbourhood exceedance probabilities.

Outside the path SURVEY section 8 names; it is what a loader's `interpolation=` argument takes (data_loaders.DataLoader).  The
reference hands the work to `xarray.DataArray.interp` (scipy.interpolate underneath, on the host).  Here (multi)linear and nearest
interpolation are what they are -- index tables and weights from the coordinates, then weighted gathers of the payload -- and run
where the payload lives: NumPy for host arrays, torch for a chunk in HBM (the tables are a few KB; the field never visits the
host).  The semantics kept from xarray / scipy:
  * targets given as plain arrays or as a 1-D DataArray over the dim itself are interpolated ORTHOGONALLY (outer product of the
    new axes); DataArray targets over other dims (stations along `index`) are interpolated POINTWISE, all such dims together;
  * `extrapolate_out_of_bounds=True` continues the outermost interval (linear) / takes the outermost value (nearest), False
    gives NaN outside the source axis;
  * a NaN corner makes the result NaN even when its weight is 0;
  * numeric non-index coordinates over interpolated dims are interpolated along (`grid_elevation`), the targets' own
    coordinates (`elevation`, station names) come with them.
"""
from __future__ import annotations

import abc
import dataclasses
from typing import Hashable, Iterable, Mapping, Optional, Sequence, Union

import numpy as np

from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import spatial
from weatherbenchx_amd.metrics import wrappers


class Interpolation(abc.ABC):
  """interpolations.py:27-58."""

  @abc.abstractmethod
  def interpolate_data_array(self, da: xr.DataArray, reference: Optional[xr.DataArray] = None) -> xr.DataArray:
    """One variable."""

  def interpolate(self, ds: Mapping[Hashable, xr.DataArray], reference: Optional[Mapping[Hashable, xr.DataArray]] = None):
    if reference is None:
      return xarray_tree.map_structure(self.interpolate_data_array, ds)
    return xarray_tree.map_structure(self.interpolate_data_array, ds, reference)


@dataclasses.dataclass
class MultipleInterpolation(Interpolation):
  """Several interpolations one after the other (interpolations.py:61-77)."""
  interpolations: Sequence[Interpolation]

  def interpolate_data_array(self, da, reference=None):
    for interpolation in self.interpolations:
      da = interpolation.interpolate_data_array(da, reference)
    return da


def pad_longitude(da: xr.DataArray) -> xr.DataArray:
  """The last meridian in front at longitude - 360 and the first behind at + 360, so that interpolation wraps
  (interpolations.py:80-86)."""
  da = xr.as_dataarray(da)
  lon = np.asarray(da['longitude'].values)
  left = da.isel(longitude=[-1]).assign_coords(longitude=lon[-1:] - 360)
  right = da.isel(longitude=[0]).assign_coords(longitude=lon[:1] + 360)
  return xr.concat([left, da, right], 'longitude')


# ---- the interpolation itself ---------------------------------------------------------------------------------------------------
def _as_float(values: np.ndarray) -> np.ndarray:
  values = np.asarray(values)
  if values.dtype.kind == 'M':
    return values.astype('datetime64[ns]').astype(np.int64).astype(np.float64)
  if values.dtype.kind == 'm':
    return values.astype('timedelta64[ns]').astype(np.int64).astype(np.float64)
  return values.astype(np.float64)


def _axis_table(source: np.ndarray, wanted: np.ndarray, method: str, extrapolate: bool):
  """For every wanted coordinate: the two source positions around it (in the source's own order), the weight of the second,
  and whether it lies outside the source axis."""
  source, wanted = _as_float(source), _as_float(wanted)
  order = np.argsort(source, kind='stable')
  axis = source[order]
  if axis.size > 1 and not np.all(np.diff(axis) > 0):
    raise ValueError('the coordinate to interpolate along has repeated values')
  outside = (wanted < axis[0]) | (wanted > axis[-1])
  if axis.size == 1:
    # an axis of one point (a chunk of a single init time, interpolated pointwise to the stations' init_time coordinate): that
    # point where it is asked for; elsewhere it is out of bounds -- constant when extrapolating, NaN otherwise
    lo = np.zeros(wanted.shape, dtype=np.int64)
    return order[lo], order[lo], np.where(np.isnan(wanted), np.nan, 0.0), outside
  lo = np.clip(np.searchsorted(axis, wanted, side='right') - 1, 0, axis.size - 2)
  weight = (wanted - axis[lo]) / (axis[lo + 1] - axis[lo])
  if method == 'nearest':
    weight = (np.clip(weight, 0.0, 1.0) > 0.5).astype(np.float64)      # a tie goes to the lower neighbour, as scipy's interpn
  elif method != 'linear':
    raise ValueError(f'unsupported interpolation method {method!r} (linear or nearest)')
  weight = np.where(np.isnan(wanted), np.nan, weight)
  return order[lo], order[lo + 1], weight, outside


def _take(data, index: np.ndarray, axis: int):
  if xr._is_torch(data):  # pylint: disable=protected-access
    import torch  # pylint: disable=g-import-not-at-top
    return data.index_select(axis, torch.as_tensor(index, device=data.device))
  return np.take(data, index, axis=axis)


def _weights_like(data, values: np.ndarray, shape):
  values = np.asarray(values, dtype=np.float64).reshape(shape)
  if xr._is_torch(data):  # pylint: disable=protected-access
    import torch  # pylint: disable=g-import-not-at-top
    return torch.as_tensor(values, device=data.device, dtype=data.dtype if data.is_floating_point() else torch.float64)
  return values


def _floating(data):
  if xr._is_torch(data):  # pylint: disable=protected-access
    return data if data.is_floating_point() else data.double()
  data = np.asarray(data)
  return data if data.dtype.kind == 'f' else data.astype(np.float64)


def _blend(a, b, w):
  """a (1 - w) + b w, written so that a NaN corner shows even under weight 0 and exact corners stay exact."""
  return a * (1 - w) + b * w


def _orthogonal(data, axis: int, table, extrapolate: bool):
  lo, hi, weight, outside = table
  shape = [1] * data.ndim
  shape[axis] = weight.size
  if not extrapolate:
    weight = np.where(outside, np.nan, weight)
  return _blend(_take(data, lo, axis), _take(data, hi, axis), _weights_like(data, weight, shape))


def _pointwise(data, axes: Sequence[int], tables, target_shape, extrapolate: bool):
  """All `axes` interpolated together at the points of the tables (each of the flattened target shape): the axes move to the
  front, collapse into one and are replaced by the target dims."""
  if xr._is_torch(data):  # pylint: disable=protected-access
    import torch  # pylint: disable=g-import-not-at-top
    moved = data.permute(*axes, *[i for i in range(data.ndim) if i not in axes])
    index = lambda a: torch.as_tensor(a, device=data.device)
  else:
    moved = np.moveaxis(data, list(axes), list(range(len(axes))))
    index = lambda a: a
  rest = tuple(moved.shape[len(axes):])
  npoint = int(np.prod(target_shape)) if len(target_shape) else 1
  total = None
  for corner in range(1 << len(axes)):
    picks, weight = [], np.ones(npoint)
    for k, (lo, hi, w, outside) in enumerate(tables):
      upper = (corner >> k) & 1
      picks.append(index((hi if upper else lo).reshape(-1)))
      w = w.reshape(-1)
      if not extrapolate:
        w = np.where(outside.reshape(-1), np.nan, w)
      weight = weight * (w if upper else 1 - w)
    part = moved[tuple(picks)] * _weights_like(data, weight, (npoint,) + (1,) * len(rest))
    total = part if total is None else total + part
  return total.reshape(tuple(target_shape) + rest)


def interp(da: xr.DataArray, dim_args: Mapping[str, Union[xr.DataArray, np.ndarray]], method: str = 'linear',
           extrapolate: bool = True) -> xr.DataArray:
  """`da.interp(**dim_args, method=method)` with the out-of-bounds behaviour chosen by `extrapolate` (module docstring)."""
  da = xr.as_dataarray(da)
  for dim in dim_args:
    if dim not in da.dims:
      raise ValueError(f'dimension {dim!r} not in {da.dims}')
  outer, points = {}, {}
  for dim, target in dim_args.items():
    if isinstance(target, xr.DataArray) and target.dims != (dim,):
      points[dim] = target
    else:
      outer[dim] = np.asarray(target.values if isinstance(target, xr.DataArray) else target)
      if outer[dim].ndim > 1:
        raise ValueError(f'target coordinates of {dim!r} as a plain array must be 0-d or 1-d')
  # non-index coordinates over the interpolated dims travel along (the reference relies on it for `grid_elevation`)
  extra = {}
  for name, (cdims, cvalues) in da._coords.items():  # pylint: disable=protected-access
    if name in da.dims or not set(cdims) & set(dim_args):
      continue
    if np.asarray(cvalues).dtype.kind in 'fiu':
      extra[name] = xr.DataArray(np.asarray(cvalues, dtype=np.float64), dims=cdims,
                                 coords={d: da.coords[d] for d in cdims if d in da.coords})
  data, dims = _floating(da.data), list(da.dims)
  coords = {k: v for k, v in da._coords.items() if not set(v[0]) & set(dim_args)}  # pylint: disable=protected-access
  for dim, wanted in outer.items():
    axis = dims.index(dim)
    table = _axis_table(da.coords[dim].values, wanted.reshape(-1), method, extrapolate)
    data = _orthogonal(data, axis, table, extrapolate)
    if wanted.ndim == 0:
      data = data.squeeze(axis)
      dims.pop(axis)
      coords[dim] = ((), wanted)
    else:
      coords[dim] = ((dim,), wanted)
  if points:
    targets = xr.broadcast(*points.values())
    tdims, tshape = targets[0].dims, targets[0].shape
    for d in tdims:
      if d in dims and d not in points:
        raise ValueError(f'target dim {d!r} collides with a dim of the array')
    axes = [dims.index(d) for d in points]
    tables = [_axis_table(da.coords[d].values, np.asarray(t.values), method, extrapolate) for d, t in zip(points, targets)]
    data = _pointwise(data, axes, tables, tshape, extrapolate)
    first = min(axes)
    kept = [d for d in dims if d not in points]
    before = sum(1 for d in kept if dims.index(d) < first)
    current = list(tdims) + kept
    dims = kept[:before] + list(tdims) + kept[before:]
    perm = [current.index(d) for d in dims]
    data = data.permute(*perm) if xr._is_torch(data) else np.transpose(data, perm)  # pylint: disable=protected-access
    for t in targets:                                                 # the targets' coordinates come along (index, elevation ...)
      for k, v in t._coords.items():  # pylint: disable=protected-access
        coords.setdefault(k, v)
    for d, t in zip(points, targets):
      coords[d] = (tuple(tdims), np.asarray(t.values))
  out = xr.DataArray._assemble(data, tuple(dims), coords, name=da.name, attrs=da.attrs)  # pylint: disable=protected-access
  for name, carried in extra.items():
    along = {d: t for d, t in dim_args.items() if d in carried.dims}
    moved = interp(carried, along, method, extrapolate)
    out._coords[name] = (tuple(moved.dims), np.asarray(moved.values))  # pylint: disable=protected-access
  return out


def interpolate_to_coords(da: xr.DataArray, dim_args: Mapping[str, Union[xr.DataArray, np.ndarray]], method: str,
                          extrapolate_out_of_bounds: bool = True) -> xr.DataArray:
  """interpolations.py:89-113."""
  return interp(da, dim_args, method, extrapolate_out_of_bounds)


def _sorted_along(da: xr.DataArray, dim: str) -> xr.DataArray:
  order = np.argsort(np.asarray(da.coords[dim].values), kind='stable')
  return da if np.array_equal(order, np.arange(order.size)) else da.isel({dim: order})


class CropToBox(Interpolation):
  """The part of the grid inside [lat_min, lat_max] x [lon_min, lon_max], after sorting both axes ascending
  (interpolations.py:116-163)."""

  def __init__(self, lat_min: float, lat_max: float, lon_min: float, lon_max: float):
    if lat_min > lat_max:
      raise ValueError(f'Invalid latitudes: {lat_min} and {lat_max}')
    if lon_min > lon_max:
      raise ValueError(f'Invalid longitudes: {lon_min} and {lon_max}')
    self._lat_min, self._lat_max, self._lon_min, self._lon_max = lat_min, lat_max, lon_min, lon_max

  def interpolate_data_array(self, da, reference=None):
    da = _sorted_along(_sorted_along(xr.as_dataarray(da), 'longitude'), 'latitude')
    return da.sel(latitude=slice(self._lat_min, self._lat_max), longitude=slice(self._lon_min, self._lon_max))


class InterpolateToFixedCoords(Interpolation):
  """To a fixed set of coordinates, longitude optionally cyclic (interpolations.py:166-211)."""

  def __init__(self, method: str, coords: Mapping[str, Union[xr.DataArray, np.ndarray]], wrap_longitude: bool = False,
               extrapolate_out_of_bounds: bool = True):
    self._method = method
    self._coords = coords
    self._wrap_longitude = wrap_longitude
    self._extrapolate_out_of_bounds = extrapolate_out_of_bounds

  def interpolate_data_array(self, da, reference=None):
    if self._wrap_longitude:
      da = pad_longitude(da)
    return interpolate_to_coords(da, self._coords, self._method, self._extrapolate_out_of_bounds)


class InterpolateToReferenceCoords(Interpolation):
  """To the coordinates of a reference array (the targets of the chunk: another grid, or stations along `index`); `dims` default
  to the array's dims that the reference has coordinates for (interpolations.py:214-293)."""

  def __init__(self, method: str, dims: Optional[Sequence[str]] = None, wrap_longitude: bool = False,
               clip_reference_coords: Optional[Iterable[str]] = None, extrapolate_out_of_bounds: bool = True):
    self._method = method
    self._dims = dims
    self._wrap_longitude = wrap_longitude
    self._clip_reference_coords = clip_reference_coords
    self._extrapolate_out_of_bounds = extrapolate_out_of_bounds

  def interpolate_data_array(self, da, reference):  # pylint: disable=arguments-renamed
    da, reference = xr.as_dataarray(da), xr.as_dataarray(reference)
    if self._wrap_longitude:
      da = pad_longitude(da)
    if self._clip_reference_coords is not None:
      for coord in self._clip_reference_coords:                       # keep the reference inside the data's extent
        source = np.asarray(da.coords[coord].values)
        cdims, cvalues = reference._coords[coord]  # pylint: disable=protected-access
        keep = (np.asarray(cvalues) >= source.min()) & (np.asarray(cvalues) <= source.max())
        reference = reference.isel({cdims[0]: np.nonzero(keep)[0]})
    dims = [d for d in da.dims if d in reference.coords] if self._dims is None else list(self._dims)
    if reference.size == 0:                                           # nothing to interpolate to: the reference's (empty) frame
      retained = [d for d in da.dims if d not in dims]
      shape = tuple(da.sizes[d] for d in retained) + tuple(reference.shape)
      coords = {k: v for k, v in reference._coords.items()}  # pylint: disable=protected-access
      coords.update({d: da._coords[d] for d in retained if d in da._coords})  # pylint: disable=protected-access
      return xr.DataArray._assemble(np.empty(shape, dtype=np.asarray(reference.values).dtype), tuple(retained) + reference.dims,  # pylint: disable=protected-access
                                    coords, name=reference.name)
    return interpolate_to_coords(da, {d: reference[d] for d in dims}, self._method, self._extrapolate_out_of_bounds)


LAPSE_RATE_K_PER_M = -0.0065  # standard atmosphere


class GridToSparseWithAltitudeAdjustment(InterpolateToReferenceCoords):
  """Grid -> stations with the height difference between station and (interpolated) model orography accounted for:
  `2m_temperature` by the standard lapse rate, `10m_wind_speed` by a factor 1 below 100 m, 1 + 0.002 (dz - 100) up to 1100 m and
  3 above; differences beyond `max_alititude_diff_in_m` (station metadata errors) are ignored.  The reference needs an
  `elevation` coordinate; other variables are only interpolated.  interpolations.py:296-397."""

  def __init__(self, method: str, grid_elevation: xr.DataArray, dims: Optional[Sequence[str]] = None, wrap_longitude: bool = False,
               extrapolate_out_of_bounds: bool = True, max_alititude_diff_in_m: float = 1500):
    self._grid_elevation = grid_elevation
    self._max_alititude_diff_in_m = max_alititude_diff_in_m
    super().__init__(method=method, dims=dims, wrap_longitude=wrap_longitude, extrapolate_out_of_bounds=extrapolate_out_of_bounds)

  def interpolate_data_array(self, da, reference):
    da = xr.as_dataarray(da)
    adjusted = da.name in ('2m_temperature', '10m_wind_speed')
    if adjusted:
      elevation = xr.as_dataarray(self._grid_elevation)
      for d in ('latitude', 'longitude'):
        np.testing.assert_allclose(np.asarray(elevation.coords[d].values), np.asarray(da.coords[d].values))
      elevation = elevation.transpose(*[d for d in da.dims if d in elevation.dims])
      da = da._replace()  # pylint: disable=protected-access
      da._coords['grid_elevation'] = (tuple(elevation.dims), np.asarray(elevation.values, dtype=np.float64))  # pylint: disable=protected-access
    out = super().interpolate_data_array(da, reference)
    if adjusted and out.size > 0:
      higher = out.coords['elevation'] - out.coords['grid_elevation']      # station above the grid box: positive
      higher = xr.DataArray(np.asarray(higher.values, dtype=np.float64), dims=higher.dims)
      higher = higher.where(abs(higher) < self._max_alititude_diff_in_m, 0)
      if da.name == '2m_temperature':
        out = out + higher * LAPSE_RATE_K_PER_M
      else:
        factor = (higher * 0 + 1).where(higher < 100, 1 + 0.002 * (higher - 100)).where(higher < 1100, 3)
        out = out * factor
      out.name = da.name
    return out


class NeighborhoodThresholdProbabilities(Interpolation):
  """A deterministic field as exceedance probabilities: the fraction of pixels of an n x n neighbourhood above each threshold (the
  FSS's fractions), for several n along `smoothing_neighborhood` (interpolations.py:400-452)."""

  def __init__(self, neighborhood_sizes, thresholds, threshold_dim='threshold_value', wrap_longitude: bool = False):
    self._neighborhood_sizes = neighborhood_sizes
    self._thresholds = thresholds
    self._threshold_dim = threshold_dim
    self._wrap_longitude = wrap_longitude

  def interpolate_data_array(self, da, reference=None):
    binary = wrappers.binarize_thresholds(da, self._thresholds, self._threshold_dim)
    parts = [spatial.neighborhood_averaging_for_single_size(binary, n, wrap_longitude=self._wrap_longitude).expand_dims(
        smoothing_neighborhood=[n]) for n in self._neighborhood_sizes]
    return xr.concat(parts, dim='smoothing_neighborhood')


class Subsample(Interpolation):
  """Every `stride`-th point along `dims` (those the array has) -- a cheaper grid without interpolating
  (interpolations.py:455-488)."""

  def __init__(self, dims: Sequence[str], stride: int):
    if stride < 1:
      raise ValueError(f'stride must be >= 1, got {stride}')
    self._dims = dims
    self._stride = stride

  def interpolate_data_array(self, da, reference=None):
    da = xr.as_dataarray(da)
    return da.isel({d: slice(None, None, self._stride) for d in self._dims if d in da.dims})
