"""Synthetic lat/lon/level/time datasets for tests, shaped like the reference's fixtures
(weatherbenchX/test_utils.py:27-90: 10 degree grid 19x36, levels 500/700/850, zeros float32 or
uniform-random float64, optional `realization` ensemble dim; dims time, latitude, longitude, level[, realization])."""
import numpy as np

from weatherbenchx_amd import xarray_lite as xr

DEFAULT_2D = ('2m_temperature',)
DEFAULT_3D = ('geopotential',)


def _times(start, stop, step):
  return np.arange(np.datetime64(start, 'ns'), np.datetime64(stop, 'ns'), np.timedelta64(step, 'h').astype('timedelta64[ns]'))


def mock_target_data(*, variables_3d=DEFAULT_3D, variables_2d=DEFAULT_2D, levels=(500, 700, 850),
                     spatial_resolution_in_degrees=10.0, time_start='2020-01-01', time_stop='2021-01-01',
                     time_resolution_hours=24, dtype=np.float32, ensemble_size=None, random=False, seed=None):
  rng = np.random.default_rng(seed)

  def val(shape):
    return rng.random(size=shape) if random else np.zeros(shape, dtype=dtype)

  nlat = round(180 / spatial_resolution_in_degrees) + 1
  nlon = round(360 / spatial_resolution_in_degrees)
  coords = {
      'time': _times(time_start, time_stop, time_resolution_hours),
      'latitude': np.linspace(-90, 90, nlat),
      'longitude': np.linspace(0, 360, nlon, endpoint=False),
      'level': np.array(levels),
  }
  if ensemble_size is not None:
    coords['realization'] = np.arange(ensemble_size)
  dims3 = tuple(coords)
  out = {}
  for name in variables_3d:
    out[name] = xr.DataArray(val(tuple(len(coords[d]) for d in dims3)), dims=dims3,
                             coords={d: coords[d] for d in dims3}, name=name)
  dims2 = tuple(d for d in dims3 if d != 'level')
  for name in variables_2d:
    out[name] = xr.DataArray(val(tuple(len(coords[d]) for d in dims2)), dims=dims2,
                             coords={d: coords[d] for d in dims2}, name=name)
  return xr.Dataset(out)


def mock_prediction_data(*, lead_start_days=0, lead_stop_days=10, lead_resolution_days=1, **kwargs):
  lead = (np.arange(lead_start_days, lead_stop_days + 1, lead_resolution_days) * 24).astype('timedelta64[h]').astype(
      'timedelta64[ns]')
  ds = mock_target_data(**kwargs)
  return ds.expand_dims(prediction_timedelta=lead)
