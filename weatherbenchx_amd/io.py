"""Persistence of metric values and AggregationStates (counterpart of WriteMetrics / WriteAggregationState and
`beam_utils.atomic_write`, weatherbenchX/beam_pipeline.py:402-443, weatherbenchX/beam_utils.py:64-101).

The reference writes netCDF through xarray; here the same variable naming (`<metric>.<variable>` for metrics,
`<statistic>#<variable>#sum_weighted_statistics|sum_weights` for states, aggregation.py:234-265) goes into a
NetCDF-3 (64-bit offset) file through `scipy.io.netcdf_file`, which xarray / netCDF4 read back directly, so the
reference's post-hoc tooling (statistical_inference, combine_results) can consume GPU-produced accumulators.
Encoding: numeric coords as they are; string coords (bin labels) as fixed-width char arrays; datetime64 /
timedelta64 coords as float64 seconds with CF `units` attributes.
"""
from __future__ import annotations

import os
import tempfile

import numpy as np

from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.aggregation import AggregationState

_EPOCH_UNITS = 'seconds since 1970-01-01 00:00:00'


def _encode_coord(values: np.ndarray):
  values = np.asarray(values)
  if values.dtype.kind == 'M':
    secs = values.astype('datetime64[ns]').astype(np.int64) / 1e9
    return secs.astype(np.float64), {'units': _EPOCH_UNITS, 'calendar': 'proleptic_gregorian'}, None
  if values.dtype.kind == 'm':
    return (values.astype('timedelta64[ns]').astype(np.int64) / 1e9).astype(np.float64), {'units': 'seconds'}, None
  if values.dtype.kind in 'US' or values.dtype == object:
    as_bytes = np.array([str(v).encode() for v in values.reshape(-1)])
    width = max(1, max(len(b) for b in as_bytes))
    chars = np.array([list(b.ljust(width).decode()) for b in as_bytes], dtype='S1').reshape(values.shape + (width,))
    return chars, {}, width
  if values.dtype == np.bool_:
    return values.astype(np.int8), {'dtype': 'bool'}, None
  if values.dtype.kind == 'i' and values.dtype.itemsize == 8:
    return values.astype(np.float64) if np.abs(values).max(initial=0) >= 2 ** 31 else values.astype(np.int32), {}, None
  return values, {}, None


def _decode_coord(var) -> np.ndarray:
  data = np.array(var[:])
  units = getattr(var, 'units', b'')
  units = units.decode() if isinstance(units, bytes) else units
  if data.dtype.kind == 'S' and data.ndim >= 1 and data.dtype.itemsize == 1:
    flat = data.reshape(-1, data.shape[-1])
    return np.array([b''.join(row).decode().rstrip() for row in flat]).reshape(data.shape[:-1])
  if units == _EPOCH_UNITS:
    return (np.round(data * 1e9).astype(np.int64)).astype('datetime64[ns]')
  if units == 'seconds':
    return (np.round(data * 1e9).astype(np.int64)).astype('timedelta64[ns]')
  if getattr(var, 'dtype_', None) is not None:
    return data
  return data


def atomic_path(path: str):
  """(temporary path, commit function): the file appears under its final name only when complete."""
  d = os.path.dirname(os.path.abspath(path)) or '.'
  os.makedirs(d, exist_ok=True)
  fd, tmp = tempfile.mkstemp(prefix='.tmp_', suffix=os.path.basename(path), dir=d)
  os.close(fd)
  return tmp, lambda: os.replace(tmp, path)


def write_dataset(dataset, path: str) -> None:
  """Mapping name -> DataArray  ->  NetCDF-3 file at `path` (atomic)."""
  import scipy.io  # pylint: disable=g-import-not-at-top
  tmp, commit = atomic_path(path)
  f = scipy.io.netcdf_file(tmp, 'w', version=2)
  try:
    dims, coords_done = {}, set()
    items = list(dataset.items())
    for _, da in items:
      for d, n in da.sizes.items():
        if dims.setdefault(d, n) != n:
          raise ValueError(f'dimension {d!r} has inconsistent sizes across variables ({dims[d]} vs {n})')
    for d, n in dims.items():
      f.createDimension(d, n)
    for _, da in items:
      for cname, (cdims, cvals) in da._coords.items():  # pylint: disable=protected-access
        if cname in coords_done or cname == 'mask':
          continue
        enc, attrs, width = _encode_coord(cvals)
        vdims = tuple(cdims)
        if width is not None:
          sdim = f'string{width}'
          if sdim not in dims:
            dims[sdim] = width
            f.createDimension(sdim, width)
          vdims = vdims + (sdim,)
        v = f.createVariable(str(cname), enc.dtype.char if enc.dtype.kind != 'S' else 'c', vdims)
        v[...] = enc
        for k, a in attrs.items():
          setattr(v, k, a)
        coords_done.add(cname)
    for name, da in items:
      vals = np.asarray(da.values, dtype=np.float64)
      v = f.createVariable(str(name), 'd', tuple(da.dims))
      if vals.ndim:
        v[...] = vals
      else:
        v.data = np.array(float(vals), dtype=np.float64)  # scalar variable (scipy's assignValue indexes a 0-d array)
      # CF: non-index coordinates (valid_time over (init_time, lead_time), a station's latitude over index) are named on the
      # variable, which is how a reader tells them from data variables
      aux = [str(c) for c in da._coords if c != 'mask' and c not in da.dims]  # pylint: disable=protected-access
      if aux:
        v.coordinates = ' '.join(aux)
  finally:
    f.close()
  commit()


def open_dataset(path: str) -> xr.Dataset:
  import scipy.io  # pylint: disable=g-import-not-at-top
  f = scipy.io.netcdf_file(path, 'r', mmap=False)
  try:
    dim_names = set(f.dimensions)
    named = set()
    for var in f.variables.values():
      listed = getattr(var, 'coordinates', b'')
      named.update((listed.decode() if isinstance(listed, bytes) else listed).split())
    coord_vars = {}
    for name, var in f.variables.items():
      vd = tuple(d for d in var.dimensions if not str(d).startswith('string'))
      if name in dim_names or name in named or (len(vd) <= 1 and var.data.dtype.kind == 'S'):
        coord_vars[name] = (vd, _decode_coord(var))
    out = {}
    for name, var in f.variables.items():
      if name in coord_vars:
        continue
      data = np.array(var[:] if var.dimensions else var.data, dtype=np.float64)
      coords = {c: (cd, cv) for c, (cd, cv) in coord_vars.items() if set(cd) <= set(var.dimensions)}
      out[name] = xr.DataArray(data, dims=tuple(var.dimensions), coords=coords, name=name, _raw_coords=True)
    return xr.Dataset(out)
  finally:
    f.close()


def write_metrics(values, out_path: str) -> None:
  """`AggregationState.metric_values(...)` -> netCDF (WriteMetrics, beam_pipeline.py:402-421)."""
  write_dataset(values, out_path)


def write_aggregation_state(state: AggregationState, out_path: str) -> None:
  """AggregationState -> netCDF with the reference's '#'-separated names (beam_pipeline.py:424-443)."""
  write_dataset(state.to_dataset(), out_path)


def read_aggregation_state(path: str) -> AggregationState:
  return AggregationState.from_dataset(open_dataset(path))
