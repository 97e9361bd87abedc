"""Chunk-level helpers the scoring path expects from the loaders (weatherbenchX/data_loaders/base.py:25-56)."""
from __future__ import annotations

from typing import Collection, Hashable, Mapping

import numpy as np

from weatherbenchx_amd import xarray_lite as xr


def add_nan_mask_to_data(data: Mapping[Hashable, xr.DataArray],
                         variable_subset: Collection[str] | None = None) -> Mapping[Hashable, xr.DataArray]:
  """Adds the boolean coordinate `mask` (True = valid, i.e. not NaN) to each variable, so that
  `Aggregator(masked=True)` skips those evaluation units (data_loaders/base.py:25-56).  For device-resident
  payloads the isnan pass runs where the data lives; the mask itself is a (small-typed) host coordinate."""
  out = dict(data)
  for name in out:
    if variable_subset is None or name in variable_subset:
      da = xr.as_dataarray(out[name])
      valid = ~da.isnull()
      da = da._replace()  # pylint: disable=protected-access
      da.coords['mask'] = xr.DataArray(np.asarray(valid.values, dtype=bool), dims=da.dims)
      out[name] = da
  return out
