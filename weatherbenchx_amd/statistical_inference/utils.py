"""Helpers shared by the inference methods (counterpart of weatherbenchX/statistical_inference/utils.py:25-138)."""
from __future__ import annotations

from typing import Any, Callable, Hashable, Sequence

import numpy as np

from weatherbenchx_amd import aggregation
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree


def get_and_check_experimental_unit_coord(aggregated_statistics: aggregation.AggregationState, name: str,
                                          check_is_dim: bool = True) -> xr.DataArray:
  """The 1-D coordinate that identifies the experimental units, the same on every statistic and variable (utils.py:25-70)."""
  coord = None
  for stat_name, stat_vars in aggregated_statistics.sum_weighted_statistics.items():
    for var_name, var in stat_vars.items():
      var = xr.as_dataarray(var)
      if name not in var.coords:
        raise ValueError(f'No experimental unit coordinate {name} found for {stat_name=} {var_name=}.')
      var_coord = var.coords[name]
      if var_coord.ndim != 1:
        raise ValueError(f'Experimental unit coordinate {name} has multiple dimensions.')
      if check_is_dim and var_coord.dims[0] != name:
        raise ValueError(f'Coordinate {name} is not a dimension coordinate.')
      if coord is None:
        coord = var_coord
      elif var_coord.size != coord.size:
        raise ValueError(f'Inconsistent sizes for coordinate {name}: {var_coord.size} and {coord.size}.')
      elif not np.all(np.asarray(var_coord.values) == np.asarray(coord.values)):
        raise ValueError(f'Inconsistent coordinate values for {name}.')
  if coord is None:
    raise ValueError('No statistics found.')
  return coord


def logarithmic_round(x, resolution=30):
  """To the nearest of `resolution` values per decade."""
  return 10 ** (np.round(np.log10(x) * resolution) / resolution)


DataArrayTree = Any


def apply_to_slices(func: Callable[..., DataArrayTree], *args: DataArrayTree, dim: Hashable | Sequence[Hashable]) -> DataArrayTree:
  """`func` on every size-1 slice of the arguments along `dim` (one dim or several: every index combination), the results joined
  again along those dims (utils.py:83-138; there through xr.combine_by_coords, here by concatenating in index order)."""
  dims = (dim,) if isinstance(dim, str) else tuple(dim)
  sizes = {}

  def check(arg):
    arg = xr.as_dataarray(arg)
    for d in dims:
      if d not in arg.dims:
        continue
      if d not in arg.coords:
        arg = arg.assign_coords({d: np.arange(arg.sizes[d])})
      if sizes.setdefault(d, arg.sizes[d]) != arg.sizes[d]:
        raise ValueError(f'Different sizes {sizes[d]}, {arg.sizes[d]} for dim={d!r}.')
    return arg

  args = xarray_tree.map_structure(check, args)
  for d in dims:
    if d not in sizes:
      raise ValueError(f'Dimension dim={d!r} not found in any arguments.')

  def along(level: int, fixed: dict):
    if level == len(dims):
      sliced = xarray_tree.map_structure(lambda a: a.isel({d: [i] for d, i in fixed.items() if d in a.dims}), args)
      return func(*sliced)
    parts = [along(level + 1, {**fixed, dims[level]: i}) for i in range(sizes[dims[level]])]
    return xarray_tree.map_structure(lambda *p: xr.concat(list(p), dim=dims[level]) if dims[level] in p[0].dims else p[0], *parts)

  return along(0, {})
