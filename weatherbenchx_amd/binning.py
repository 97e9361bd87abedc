"""Binning plugins (counterpart of weatherbenchX/binning.py:22-201: Binning, Regions, LandSea).

`create_bin_mask(statistic)` returns a boolean DataArray [bin, latitude, longitude]; the masks are
plan-time objects (O(bins*lat*lon) booleans) that are folded into W and multiplied inside the stage-2
HIP contraction.  Bounds are inclusive, longitudes are compared modulo 360 with wrap-around
(binning.py:52-89).  The coordinate / time binnings (binning.py:204-705: LatitudeBins, LongitudeBins, ByExactCoord,
ByTimeUnit, ByTimeUnitSets, ByTimeUnitFromSeconds, ByCoordBins, BySets) build their masks from coordinates alone with NumPy --
a few booleans per bin and coordinate value -- and reach the device through the same W as every other mask.
"""
from __future__ import annotations

import abc
from typing import Any, Hashable, Mapping, Optional, Sequence, Tuple

import numpy as np

from weatherbenchx_amd import xarray_lite as xr


class Binning(abc.ABC):
  """Base class: `bin_dim_name` + `create_bin_mask` (binning.py:22-49)."""

  def __init__(self, bin_dim_name: str):
    self.bin_dim_name = bin_dim_name

  @abc.abstractmethod
  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    ...


def _lat_mask(lat: xr.DataArray, lims) -> xr.DataArray:
  lo, hi = lims
  if lo >= hi:
    raise ValueError(f'`lat_lims[0]` must be smaller than `lat_lims[1]`, got {lims}`')
  return (lat >= lo) & (lat <= hi)


def _lon_mask(lon: xr.DataArray, lims) -> xr.DataArray:
  lon = lon % 360
  lo, hi = np.mod(lims[0], 360), np.mod(lims[1], 360)
  if hi > lo:
    return (lon >= lo) & (lon <= hi)
  return (lon <= hi) | (lon >= lo)  # wraps around the date line (also (0, 360) -> everything)


def _region_to_mask(lat, lon, lat_lims, lon_lims) -> xr.DataArray:
  return _lat_mask(lat, lat_lims) & _lon_mask(lon, lon_lims)


class Regions(Binning):
  """Rectangular lat/lon regions, optionally doubled with `{name}_land` bins (binning.py:147-201)."""

  def __init__(self, regions: Mapping[Hashable, Tuple[Tuple[float, float], Tuple[float, float]]],
               bin_dim_name: str = 'region', land_sea_mask: Optional[xr.DataArray] = None):
    super().__init__(bin_dim_name)
    self._regions = regions
    self._land_sea_mask = land_sea_mask

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    lat, lon = statistic['latitude'], statistic['longitude']
    per_region = []
    for name, (lat_lims, lon_lims) in self._regions.items():
      m = _region_to_mask(lat, lon, lat_lims, lon_lims).expand_dims({self.bin_dim_name: np.array([name])})
      per_region.append(m)
    masks = xr.concat(per_region, dim=self.bin_dim_name)
    if self._land_sea_mask is not None:
      lsm = xr.as_dataarray(self._land_sea_mask)
      same = (np.array_equal(np.sort(masks['latitude'].values), np.sort(lsm['latitude'].values))
              and np.array_equal(masks['longitude'].values, lsm['longitude'].values))
      assert same, 'Land/sea mask coordinates do not match.'
      land = masks & lsm.astype(bool)
      names = np.array([f'{r}_land' for r in masks[self.bin_dim_name].values])
      land = land.assign_coords({self.bin_dim_name: names}).transpose(*masks.dims)
      masks = xr.concat([masks, land], dim=self.bin_dim_name)
    return masks


class LandSea(Binning):
  """['land', 'sea'(, 'global')] masks from a land fraction field (binning.py:92-144)."""

  def __init__(self, land_sea_fraction: xr.DataArray, land_sea_threshold: float = 0.5,
               bin_dim_name: str = 'land_sea', include_global_mask: bool = False):
    super().__init__(bin_dim_name)
    self._land_mask = xr.as_dataarray(land_sea_fraction) >= land_sea_threshold
    self._include_global_mask = include_global_mask

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    land = self._land_mask
    layers, labels = [land, ~land], ['land', 'sea']
    if self._include_global_mask:
      layers.append(xr.ones_like(land, dtype=bool))
      labels.append('global')
    stacked = xr.concat([m.expand_dims(self.bin_dim_name) for m in layers], dim=self.bin_dim_name)
    return stacked.assign_coords({self.bin_dim_name: np.array(labels)})


# ---- coordinate / time binnings (binning.py:204-705) -----------------------------------------------------------------------------
def _coordinate(statistic: xr.DataArray, name: str):
  """(values, dims) of a dimension or non-dimension coordinate of the statistic."""
  statistic = xr.as_dataarray(statistic)
  if name not in statistic.coords:
    raise KeyError(f'the statistic has no coordinate {name!r} (coordinates: {list(statistic.coords)})')
  c = statistic.coords[name]
  return np.asarray(c.values), tuple(c.dims)


def _mask_array(statistic, bin_dim_name, labels, mask, dims):
  """bool DataArray [bin, *dims] with the statistic's coordinates on `dims` (the frame the Aggregator multiplies into W)."""
  statistic = xr.as_dataarray(statistic)
  coords = {bin_dim_name: np.asarray(labels)}
  for d in dims:
    if d in statistic.coords and d != bin_dim_name:
      coords[d] = statistic.coords[d]
  return xr.DataArray(np.asarray(mask, dtype=bool), dims=(bin_dim_name,) + tuple(dims), coords=coords)


def _with_global(labels, mask, first: bool):
  """One more bin that holds everything.  Bin labels of one coordinate share a dtype: strings all round when they differ."""
  labels = np.asarray(labels)
  everything = np.ones((1,) + mask.shape[1:], dtype=bool)
  if labels.dtype.kind not in 'US':
    labels = labels.astype(str)
  g = np.array(['global'])
  return ((np.concatenate([g, labels]), np.concatenate([everything, mask])) if first
          else (np.concatenate([labels, g]), np.concatenate([mask, everything])))


def vectorized_coord_mask(coord: xr.DataArray, coord_name: str, bin_dim_name: str, add_global_bin: bool = False) -> xr.DataArray:
  """One bin per distinct value of `coord` (binning.py:291-322); works for an empty statistic too."""
  coord = xr.as_dataarray(coord)
  values = np.asarray(coord.values)
  unique = np.unique(values)
  mask = values[None, ...] == unique.reshape((-1,) + (1,) * values.ndim)
  labels = unique
  if add_global_bin:
    labels, mask = _with_global(labels, mask, first=True)
  coords = {bin_dim_name: labels}
  for d in coord.dims:
    if d in coord.coords and d != bin_dim_name and d != coord_name:
      coords[d] = coord.coords[d]
  return xr.DataArray(mask, dims=(bin_dim_name,) + tuple(coord.dims), coords=coords)


class LatitudeBins(Binning):
  """Latitude bands of `degrees` width: [start, start + degrees], inclusive like every latitude mask here (binning.py:204-243)."""

  def __init__(self, degrees: float, lat_range: Tuple[int, int] = (-90, 90), bin_dim_name: str = 'latitude_bins'):
    super().__init__(bin_dim_name)
    self._degrees = degrees
    self._starts = np.arange(lat_range[0], lat_range[1] + degrees, degrees)[:-1]

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    statistic = xr.as_dataarray(statistic)
    lat = statistic['latitude']
    bands = np.stack([np.asarray(_lat_mask(lat, (a, a + self._degrees)).values) for a in self._starts])
    idx = (slice(None),) + tuple(slice(None) if d in lat.dims else None for d in statistic.dims)  # (latitude may be a coordinate over `index`)
    mask = np.broadcast_to(bands[idx], (len(self._starts),) + tuple(statistic.shape))  # broadcast to the statistic, as the reference
    return _mask_array(statistic, self.bin_dim_name, self._starts, mask, statistic.dims)


class LongitudeBins(Binning):
  """Longitude bands of `degrees` width, compared modulo 360, labelled by their start modulo 360 (binning.py:246-288)."""

  def __init__(self, degrees: float, lon_range: Tuple[int, int] = (0, 360), bin_dim_name: str = 'longitude_bins'):
    super().__init__(bin_dim_name)
    self._degrees = degrees
    end = lon_range[1] + (360 if lon_range[0] >= lon_range[1] else 0)
    self._starts = np.arange(lon_range[0], end + degrees, degrees)[:-1]

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    statistic = xr.as_dataarray(statistic)
    lon = statistic['longitude']
    bands = np.stack([np.asarray(_lon_mask(lon, (a, a + self._degrees)).values) for a in self._starts])
    idx = (slice(None),) + tuple(slice(None) if d in lon.dims else None for d in statistic.dims)  # (longitude may be a coordinate over `index`)
    mask = np.broadcast_to(bands[idx], (len(self._starts),) + tuple(statistic.shape))
    return _mask_array(statistic, self.bin_dim_name, np.mod(self._starts, 360), mask, statistic.dims)


class ByExactCoord(Binning):
  """One bin per distinct value of a NON-dimension coordinate, e.g. the lead time of sparse forecasts (binning.py:325-358)."""

  def __init__(self, coord: str, add_global_bin: bool = False):
    super().__init__(coord)
    self.coord = coord
    self.add_global_bin = add_global_bin

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    statistic = xr.as_dataarray(statistic)
    assert self.coord not in statistic.dims, 'For dimensions, specify reduce_dims in aggregation.'
    return vectorized_coord_mask(statistic.coords[self.coord], self.coord, self.coord, self.add_global_bin)


_SECONDS = {'second': 1, 'minute': 60, 'hour': 3600, 'day': 86400, 'week': 7 * 86400, 'year': 365 * 86400}


def _extract_time_unit(values: np.ndarray, unit: str) -> np.ndarray:
  """The `.dt` field `unit` of datetime64 values, or whole `unit`s of timedelta64 values (binning.py:361-395)."""
  values = np.asarray(values)
  if values.dtype.kind == 'm':
    if unit not in _SECONDS:
      raise ValueError(f'Unsupported unit for timedelta: {unit}')
    seconds = values.astype('timedelta64[ns]').astype(np.int64) / 1e9
    return seconds // _SECONDS[unit] if unit != 'second' else seconds
  if values.dtype.kind != 'M':
    raise TypeError(f'a datetime64 or timedelta64 coordinate is needed to bin by {unit!r} (got {values.dtype})')
  t = values.astype('datetime64[ns]')
  day = t.astype('datetime64[D]')
  if unit == 'year':
    return t.astype('datetime64[Y]').astype(np.int64) + 1970
  if unit == 'month':
    return t.astype('datetime64[M]').astype(np.int64) % 12 + 1
  if unit == 'day':
    return (day - t.astype('datetime64[M]').astype('datetime64[D]')).astype(np.int64) + 1
  if unit == 'dayofyear':
    return (day - t.astype('datetime64[Y]').astype('datetime64[D]')).astype(np.int64) + 1
  if unit in ('dayofweek', 'weekday'):
    return (day.astype(np.int64) + 3) % 7  # 1970-01-01 was a Thursday; Monday = 0
  if unit == 'hour':
    return (t - day).astype('timedelta64[h]').astype(np.int64)
  if unit == 'minute':
    return (t - t.astype('datetime64[h]')).astype('timedelta64[m]').astype(np.int64)
  if unit == 'second':
    return (t - t.astype('datetime64[m]')).astype('timedelta64[s]').astype(np.int64)
  # the other integer fields of xarray's `.dt` accessor the reference forwards to (binning.py: `getattr(coord.dt, unit)`)
  month = t.astype('datetime64[M]').astype(np.int64) % 12 + 1
  if unit == 'quarter':
    return (month - 1) // 3 + 1
  if unit in ('days_in_month', 'daysinmonth'):
    first = t.astype('datetime64[M]')
    return ((first + 1).astype('datetime64[D]') - first.astype('datetime64[D]')).astype(np.int64)
  if unit in ('week', 'weekofyear'):  # ISO 8601 week number: the week of the year that holds this date's Thursday
    thursday = day + ((3 - (day.astype(np.int64) + 3) % 7).astype('timedelta64[D]'))
    return (thursday - thursday.astype('datetime64[Y]').astype('datetime64[D]')).astype(np.int64) // 7 + 1
  if unit == 'is_leap_year':
    y = t.astype('datetime64[Y]').astype(np.int64) + 1970
    return ((y % 4 == 0) & ((y % 100 != 0) | (y % 400 == 0))).astype(np.int64)
  if unit == 'microsecond':
    return (t - t.astype('datetime64[s]')).astype('timedelta64[us]').astype(np.int64)
  if unit == 'nanosecond':
    return (t - t.astype('datetime64[us]')).astype('timedelta64[ns]').astype(np.int64)
  raise ValueError(f'Unsupported unit for datetime: {unit} (supported: year, quarter, month, week / weekofyear, day, dayofyear, '
                   'dayofweek / weekday, days_in_month, is_leap_year, hour, minute, second, microsecond, nanosecond; `season` is '
                   'a string field: bin by month sets with ByTimeUnitSets)')


class ByTimeUnit(Binning):
  """One bin per value of a time unit along a datetime64 / timedelta64 coordinate: all initialisations at the same hour of the
  day, all lead times within the same day ... (binning.py:398-441)."""

  def __init__(self, unit: str, time_dim: str, add_global_bin: bool = False):
    super().__init__(f'{time_dim}_{unit}')
    self.unit = unit
    self.time_dim = time_dim
    self.add_global_bin = add_global_bin

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    values, dims = _coordinate(statistic, self.time_dim)
    field = _extract_time_unit(values, self.unit)
    unique = np.unique(field)
    mask = field[None, ...] == unique.reshape((-1,) + (1,) * field.ndim)
    labels = unique
    if self.add_global_bin:
      labels, mask = _with_global(labels, mask, first=True)
    return _mask_array(statistic, self.bin_dim_name, labels, mask, dims)


def _as_set(s) -> np.ndarray:
  return np.array(list(s) if (isinstance(s, (Sequence, np.ndarray)) and not isinstance(s, str)) else [s])


def _isin(values: np.ndarray, members: np.ndarray) -> np.ndarray:
  if members.size == 0:
    return np.zeros(values.shape, dtype=bool)
  try:
    return np.isin(values, members)
  except (TypeError, ValueError):  # (values that cannot be compared with the coordinate are simply not in it)
    return np.zeros(values.shape, dtype=bool)


class ByTimeUnitSets(Binning):
  """Named SETS of time-unit values: {'00/12': [0, 12], '06/18': [6, 18]} of `hour` along `init_time` (binning.py:444-518)."""

  def __init__(self, sets: Mapping[str, Any], unit: str, dim: str, bin_dim_name: Optional[str] = None, add_global_bin: bool = False):
    super().__init__(bin_dim_name if bin_dim_name is not None else f'{dim}_{unit}_sets')
    self.sets = sets
    self.unit = unit
    self.dim = dim
    self.add_global_bin = add_global_bin

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    values, dims = _coordinate(statistic, self.dim)
    field = _extract_time_unit(values, self.unit)
    labels = np.array([str(k) for k in self.sets])
    mask = np.stack([_isin(field, _as_set(s)) for s in self.sets.values()]) if len(self.sets) else np.zeros((0,) + field.shape, bool)
    if self.add_global_bin:
      labels, mask = _with_global(labels, mask, first=False)
    return _mask_array(statistic, self.bin_dim_name, labels, mask, dims)


class ByTimeUnitFromSeconds(Binning):
  """ByTimeUnit for a coordinate that already holds plain seconds; `bins` default to a clock face (binning.py:521-569)."""

  def __init__(self, unit: str, time_dim: str, bins: Optional[Sequence[int]] = None):
    super().__init__(f'{time_dim}_{unit}')
    self.unit = unit
    self.time_dim = time_dim
    self.bins = bins

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    values, dims = _coordinate(statistic, self.time_dim)
    if self.unit not in ('second', 'minute', 'hour'):
      raise ValueError(f'Unsupported unit: {self.unit}')
    field = np.asarray(values) // {'second': 1, 'minute': 60, 'hour': 3600}[self.unit]
    bins = np.asarray(self.bins if self.bins is not None else np.arange(0, 24 if self.unit == 'hour' else 60))
    mask = field[None, ...] == bins.reshape((-1,) + (1,) * field.ndim)
    return _mask_array(statistic, self.bin_dim_name, bins, mask, dims)


class ByCoordBins(Binning):
  """Half-open bins [edge_i, edge_i+1) over a coordinate; the bin dim takes the coordinate's name and the left edges as labels
  (binning.py:572-637)."""

  def __init__(self, dim_name: str, bin_edges: np.ndarray, add_global_bin: bool = False):
    super().__init__(dim_name)
    self.dim_name = dim_name
    self.bin_edges = bin_edges
    self.add_global_bin = add_global_bin

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    statistic = xr.as_dataarray(statistic)
    values, dims = _coordinate(statistic, self.dim_name)
    edges = np.asarray(self.bin_edges)
    starts, stops = edges[:-1], edges[1:]
    mask = np.stack([(values >= a) & (values < b) for a, b in zip(starts, stops)]) if len(starts) else np.zeros((0,) + values.shape, bool)
    labels = starts
    if self.add_global_bin:
      # `str(start)` per edge (binning.py:605): '1 hours' for a timedelta64 edge, where `.astype(str)` would say '1'
      labels, mask = _with_global(np.array([str(a) for a in starts], dtype=str), mask, first=False)
    if dims == (self.dim_name,):
      # binning a DIMENSION by its own name: the mask keeps that dim under a private name so that [bin, dim] stays a matrix
      # (the reference drops the coordinate and reuses the name for the bin dim, which only works for non-dimension coordinates)
      raise ValueError(f'ByCoordBins({self.dim_name!r}): {self.dim_name!r} is a dimension of the statistic; bin a dimension with a '
                       'coordinate of another name (assign_coords) or reduce over it')
    return _mask_array(statistic, self.dim_name, labels, mask, dims)


class BySets(Binning):
  """Named sets of coordinate values (station names ...), optionally each set's complement and a global bin (binning.py:640-705)."""

  def __init__(self, sets: Mapping[str, Any], coord_name: str, bin_dim_name: Optional[str] = None, add_set_complements: bool = False,
               add_global_bin: bool = False):
    if bin_dim_name is None or bin_dim_name == coord_name:
      raise ValueError('bin_dim_name must be defined and be different from coord_name.')
    super().__init__(bin_dim_name)
    self.sets = sets
    self.coord_name = coord_name
    self.add_set_complements = add_set_complements
    self.add_global_bin = add_global_bin

  def create_bin_mask(self, statistic: xr.DataArray) -> xr.DataArray:
    values, dims = _coordinate(statistic, self.coord_name)
    labels, layers = [], []
    for name, s in self.sets.items():
      inside = _isin(values, _as_set(s))
      labels.append(str(name))
      layers.append(inside)
      if self.add_set_complements:
        labels.append(f'not_in_{name}')
        layers.append(~inside)
    labels = np.array(labels)
    mask = np.stack(layers) if layers else np.zeros((0,) + values.shape, bool)
    if self.add_global_bin:
      labels, mask = _with_global(labels, mask, first=False)
    return _mask_array(statistic, self.bin_dim_name, labels, mask, dims)
