"""Weighting plugins (counterpart of weatherbenchX/weighting.py:24-130; GridAreaWeighting only --
StationDensityWeighting is the sparse-observation path, out of scope per SURVEY section 2).

The weights are a float64 vector over latitude (plan-time, O(nlat)); the per-point multiply happens in
the stage-2 HIP contraction (csrc/wbx_s2.hip).  NOT cos(lat): exact cell areas sin(ub) - sin(lb) with
mid-point bounds clipped to the poles (weighting.py:62-88).
"""
from __future__ import annotations

import abc
import dataclasses

import numpy as np

from weatherbenchx_amd import xarray_lite as xr


class Weighting(abc.ABC):
  """`weights(statistic)` returns a DataArray broadcastable against the statistic (weighting.py:24-42)."""

  @abc.abstractmethod
  def weights(self, statistic: xr.DataArray) -> xr.DataArray:
    ...


def _is_strictly_monotonic(v) -> bool:
  d = np.diff(v)
  return bool(np.all(d > 0) or np.all(d < 0))


def latitude_cell_bounds(x: np.ndarray) -> np.ndarray:
  """Cell edges for increasing cell centres in radians: mid-points, end cells extended by half a spacing
  and clipped to [-pi/2, pi/2] (weighting.py:62-79)."""
  if not np.all(np.diff(x) > 0):
    raise AssertionError('Points must be increasing.')
  half_pi = np.pi / 2
  first = max(x[0] - (x[1] - x[0]) / 2, -half_pi)
  last = min(x[-1] + (x[-1] - x[-2]) / 2, half_pi)
  edges = np.empty(len(x) + 1, dtype=x.dtype)
  edges[0], edges[-1] = first, last
  edges[1:-1] = (x[:-1] + x[1:]) / 2
  return edges


def cell_area_from_latitude(points: np.ndarray) -> np.ndarray:
  """Integral of cos(lat) over each cell (weighting.py:82-88)."""
  edges = latitude_cell_bounds(points)
  return np.sin(edges[1:]) - np.sin(edges[:-1])


@dataclasses.dataclass
class GridAreaWeighting(Weighting):
  """Area of rectangular lat/lon cells, optionally normalised to mean 1 (weighting.py:91-130)."""

  latitude_name: str = 'latitude'
  return_normalized: bool = True

  def weights(self, statistic: xr.DataArray) -> xr.DataArray:
    if self.latitude_name not in statistic.dims:
      return xr.DataArray(1)
    lat_coord = statistic[self.latitude_name]
    lat = np.asarray(lat_coord.values)
    assert _is_strictly_monotonic(lat), f'Points must be strictly monotonic: {lat}'
    descending = lat[0] > lat[1]
    ordered = lat[::-1] if descending else lat
    w = cell_area_from_latitude(np.deg2rad(ordered))
    if descending:
      w = w[::-1]
    if self.return_normalized:
      w = w / np.mean(w)
    return lat_coord.copy(data=np.ascontiguousarray(w))
