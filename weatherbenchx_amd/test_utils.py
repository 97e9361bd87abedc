"""Mock datasets with the reference's helper signatures (counterpart of weatherbenchX/test_utils.py:27-104), as labelled arrays of
this package: a regular latitude / longitude grid with both poles, pressure levels, a `time` axis, optionally a `realization`
ensemble dim; zeros of `dtype`, or uniform random float64 with `random=True`.  Dims come in the order time, latitude, longitude,
level[, realization] (the reference builds the 2-D variables' dims from a set, i.e. in no particular order)."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import pandas as pd

from weatherbenchx_amd import xarray_lite as xr

DEFAULT_2D_VARIABLES = ('2m_temperature',)
DEFAULT_3D_VARIABLES = ('geopotential',)


def _ns(delta) -> np.timedelta64:
  return (pd.Timedelta(delta) if isinstance(delta, str) else pd.Timedelta(delta)).to_timedelta64().astype('timedelta64[ns]')


def mock_target_data(*, variables_3d: Sequence[str] = DEFAULT_3D_VARIABLES, variables_2d: Sequence[str] = DEFAULT_2D_VARIABLES,
                     levels: Sequence[int] = (500, 700, 850), spatial_resolution_in_degrees: float = 10.0, time_start: str = '2020-01-01',
                     time_stop: str = '2021-01-01', time_resolution='1 day', dtype=np.float32, ensemble_size: Optional[int] = None,
                     random: bool = False, seed: Optional[int] = None) -> xr.Dataset:
  """An analysis-like dataset over [time_start, time_stop) (test_utils.py:27-82)."""
  rng = np.random.default_rng(seed)
  fill = (lambda shape: rng.random(size=shape)) if random else (lambda shape: np.zeros(shape, dtype=dtype))
  coords = {
      'time': np.arange(np.datetime64(time_start, 'ns'), np.datetime64(time_stop, 'ns'), _ns(time_resolution)),
      'latitude': np.linspace(-90, 90, round(180 / spatial_resolution_in_degrees) + 1),
      'longitude': np.linspace(0, 360, round(360 / spatial_resolution_in_degrees), endpoint=False),
      'level': np.array(levels),
  }
  if ensemble_size is not None:
    coords['realization'] = np.arange(ensemble_size)
  out = {}
  for names, dims in ((variables_3d, tuple(coords)), (variables_2d, tuple(d for d in coords if d != 'level'))):
    for name in names:
      out[name] = xr.DataArray(fill(tuple(len(coords[d]) for d in dims)), dims=dims, coords={d: coords[d] for d in dims}, name=name)
  return xr.Dataset(out)


def mock_prediction_data(*, lead_start='0 day', lead_stop='10 day', lead_resolution='1 day', **kwargs) -> xr.Dataset:
  """The same with a leading `prediction_timedelta` dim from lead_start to lead_stop INCLUSIVE (test_utils.py:85-104)."""
  first, last, step = _ns(lead_start), _ns(lead_stop), _ns(lead_resolution)
  leads = first + step * np.arange(int((last - first) // step) + 1)
  return mock_target_data(**kwargs).expand_dims(prediction_timedelta=leads)
