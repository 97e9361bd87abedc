"""Binning plugins (counterpart of weatherbenchX/binning.py:22-201: Binning, Regions, LandSea).

`create_bin_mask(statistic)` returns a boolean DataArray [bin, latitude, longitude]; the masks are
plan-time objects (O(bins*lat*lon) booleans) that are folded into W and multiplied inside the stage-2
HIP contraction.  Bounds are inclusive, longitudes are compared modulo 360 with wrap-around
(binning.py:52-89).  The time/coordinate binnings (binning.py:204-705) are out of scope.
"""
from __future__ import annotations

import abc
from typing import Hashable, Mapping, Optional, Tuple

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
