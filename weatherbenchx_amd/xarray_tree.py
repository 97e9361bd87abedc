"""map_structure over nested containers of DataArrays (counterpart of weatherbenchX/xarray_tree.py:42-68).

Leaves are DataArrays (or anything non-container).  A `Dataset` maps variable-wise: `None` results are
dropped and the result is a Dataset again when every result is a DataArray with compatible coordinates,
otherwise a plain dict -- the behaviour `PerVariableStatisticWithClimatology.compute` and
`AggregationState.sum` rely on.
"""
from __future__ import annotations

from typing import Any, Callable

from weatherbenchx_amd import xarray_lite as xr


def map_structure(func: Callable[..., Any], *structures: Any) -> Any:
  if not callable(func):
    raise TypeError(f'func must be callable, got: {func}')
  if not structures:
    raise ValueError('Must provide at least one structure')
  first = structures[0]
  if isinstance(first, xr.Dataset):
    results = {name: func(*[s[name] for s in structures]) for name in first.keys()}
    if all(r is None or isinstance(r, xr.DataArray) for r in results.values()):
      kept = {k: v for k, v in results.items() if v is not None}
      if _coords_compatible(kept.values()):
        return xr.Dataset(kept)
    return results
  if isinstance(first, dict):
    return {name: map_structure(func, *[s[name] for s in structures]) for name in first.keys()}
  if isinstance(first, (list, tuple, set)):
    return type(first)(map_structure(func, *group) for group in zip(*structures))
  return func(*structures)


def _coords_compatible(arrays) -> bool:
  seen = {}
  for a in arrays:
    for name, (dims, vals) in a._coords.items():  # pylint: disable=protected-access
      if name in seen:
        d0, v0 = seen[name]
        if d0 != dims or not xr._values_equal(v0, vals):  # pylint: disable=protected-access
          return False
      else:
        seen[name] = (dims, vals)
  return True
