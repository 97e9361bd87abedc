"""Zonal power / energy spectra: batched rocFFT along longitude + HIP |F|^2 reduction.

There is NO spectrum metric (and no test) in the reference snapshot, so this component has
"parity unpinned"; the definition follows the WeatherBench-2 lineage the reference's README points to:
    F_k = rfft(f along longitude) / nlon,   S_k = |F_k|^2 * (1 if k == 0 else 2),   k = 0 .. nlon // 2
optionally times the circle of latitude C(lat) = 2 pi R cos(lat) ("energy" per unit wavenumber).  It is pinned
by analytic tests (Parseval, constant field, single sinusoid) and by the float64 `numpy.fft.rfft` oracle.

`ZonalPowerSpectrum` plugs into the same Statistic / Aggregator protocol: the statistic's dims are the field's
dims with `longitude` replaced by `zonal_wavenumber`; aggregating it over time / latitude (area weights) is ONE
fused launch per tile of rows -- the per-row spectra are never materialised.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import engine
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base

EARTH_RADIUS_M = 6371.0e3


def _run_spectrum(field: xr.DataArray, lon_dim: str, group: np.ndarray, scale: np.ndarray, ngroup: int) -> np.ndarray:
  """power[ngroup][nk]; rows = all non-longitude dims in the field's own order."""
  ctx = _hip.default_context()
  data = field.data
  engine._sync_torch_producers([data])  # pylint: disable=protected-access
  dev = engine._to_device(ctx, field, _hip.F32)  # pylint: disable=protected-access
  if engine._common_dtype([data]) != _hip.F32:  # pylint: disable=protected-access
    raise TypeError('zonal spectra take float32 fields (rocFFT single precision); cast the input')
  nlon = field.sizes[lon_dim]
  nk = nlon // 2 + 1
  row_dims = [d for d in field.dims if d != lon_dim]
  lon_stride = dev.layout.stride(lon_dim)
  # rows must form ONE uniformly strided batch for rocFFT: collapse the row dims if their strides nest,
  # otherwise loop over the outer dims.
  sizes = [field.sizes[d] for d in row_dims]
  strides = [dev.layout.stride(d) for d in row_dims]
  order = np.argsort(strides)[::-1] if row_dims else []
  # find the largest suffix (in stride order) that is uniformly nested: stride[i] == stride[i+1] * size[i+1]
  inner = []
  for idx in order[::-1]:
    if not inner or strides[idx] == strides[inner[-1]] * sizes[inner[-1]]:
      inner.append(idx)
    else:
      break
  inner_set = set(inner)
  outer = [i for i in range(len(row_dims)) if i not in inner_set]
  batch = int(np.prod([sizes[i] for i in inner], dtype=np.int64)) if inner else 1
  row_stride = strides[inner[0]] if inner else 1
  g_dev = ctx.upload(np.ascontiguousarray(group, dtype=np.int32))
  s_dev = ctx.upload(np.ascontiguousarray(scale, dtype=np.float64))
  out = ctx.alloc(max(ngroup * nk, 1) * 8)
  # row index (C order over row_dims) of each (outer combo, inner position)
  row_index = np.arange(int(np.prod(sizes, dtype=np.int64)) if sizes else 1, dtype=np.int64).reshape(sizes or [1])
  inner_by_stride_desc = sorted(inner, key=lambda i: -strides[i])
  perm = outer + inner_by_stride_desc
  row_index = np.transpose(row_index, perm).reshape(-1, batch) if row_dims else row_index.reshape(1, 1)
  outer_offsets = np.zeros((), dtype=np.int64)
  for i in outer:
    outer_offsets = outer_offsets[..., None] + np.arange(sizes[i], dtype=np.int64) * strides[i]
  outer_offsets = np.asarray(outer_offsets).reshape(-1)
  first = True
  for o, base_off in enumerate(outer_offsets):
    rows = row_index[o]
    contiguous_rows = bool(np.all(np.diff(rows) == 1)) if rows.size > 1 else True
    if contiguous_rows:
      g_ptr, s_ptr = g_dev.ptr + 4 * int(rows[0]), s_dev.ptr + 8 * int(rows[0])
      keep = None
    else:  # the batch walks the rows in another order than C order: permute group/scale for this slab
      keep = (ctx.upload(np.ascontiguousarray(group[rows], dtype=np.int32)),
              ctx.upload(np.ascontiguousarray(scale[rows], dtype=np.float64)))
      g_ptr, s_ptr = keep[0].ptr, keep[1].ptr
    _hip.check(ctx.lib.wbx_zonal_spectrum(ctx.handle, C.c_void_p(dev.ptr + 4 * int(base_off)), int(lon_stride),
                                          int(row_stride), int(batch), int(nlon), C.c_void_p(g_ptr), C.c_void_p(s_ptr),
                                          int(ngroup), 0 if first else 1, C.c_void_p(out.ptr)), 'wbx_zonal_spectrum')
    first = False
    if keep is not None:
      ctx.synchronize()
  return ctx.download(out.ptr, (ngroup, nk), np.float64)


class LazySpectrum(xr.DataArray):
  """Per-row zonal spectrum of a field: a DataArray whose payload is only computed on demand."""

  def __init__(self, source: xr.DataArray, lon_dim: str, k_dim: str, circumference: bool, lat_dim: str):
    self._data = None
    nlon = source.sizes[lon_dim]
    self._nk = nlon // 2 + 1
    self._dims = tuple(k_dim if d == lon_dim else d for d in source.dims)
    self.name = source.name
    self.attrs = {}
    self._coords = {k: v for k, v in source._coords.items() if lon_dim not in v[0]}  # pylint: disable=protected-access
    self._coords[k_dim] = ((k_dim,), np.arange(self._nk))
    self._source, self._lon_dim, self._k_dim = source, lon_dim, k_dim
    self._circumference, self._lat_dim = circumference, lat_dim

  @property
  def is_lazy(self):
    return self._data is None

  @property
  def shape(self):
    return tuple(self._nk if d == self._k_dim else self._source.sizes[d] for d in self._dims)

  @property
  def dtype(self):
    return np.dtype(np.float64)

  def row_scale(self) -> xr.DataArray:
    """Per-row factor that belongs to the statistic itself (C(lat) for the energy spectrum)."""
    if not self._circumference:
      return xr.DataArray(np.float64(1.0))
    lat = self._source[self._lat_dim]
    return lat.copy(data=2 * np.pi * EARTH_RADIUS_M * np.cos(np.deg2rad(np.asarray(lat.values, dtype=np.float64))))

  def reduce_rows(self, row_weight: xr.DataArray, kept_dims) -> np.ndarray:
    """sum over the non-kept row dims of weight * spectrum -> array over (kept_dims..., k)."""
    row_dims = [d for d in self._source.dims if d != self._lon_dim]
    sizes = {d: self._source.sizes[d] for d in row_dims}
    shape = [sizes[d] for d in row_dims]
    w = (row_weight * self.row_scale()).astype(np.float64)
    scale = np.broadcast_to(xr._bcast_data(w, row_dims, sizes), shape).reshape(-1)  # pylint: disable=protected-access
    kept = [d for d in row_dims if d in kept_dims]
    g = np.zeros((), dtype=np.int64)
    mult = 1
    mults = {}
    for d in reversed(kept):
      mults[d] = mult
      mult *= sizes[d]
    for d in row_dims:
      g = g[..., None] + np.arange(sizes[d], dtype=np.int64) * mults.get(d, 0)
    group = np.broadcast_to(g, shape).reshape(-1)
    out = _run_spectrum(self._source, self._lon_dim, group, scale, int(mult))
    return out.reshape([sizes[d] for d in kept] + [self._nk]), tuple(kept) + (self._k_dim,)

  @property
  def data(self):
    if self._data is None:
      row_dims = [d for d in self._source.dims if d != self._lon_dim]
      arr, dims = self.reduce_rows(xr.DataArray(np.float64(1.0)), row_dims)
      self._data = np.transpose(arr, [dims.index(d) for d in self._dims])
    return self._data


class ZonalPowerSpectrum(base.PerVariableStatistic):
  """S_k of predictions or targets along longitude (see module docstring).  No reference counterpart."""

  def __init__(self, which: str = 'predictions', longitude_name: str = 'longitude', latitude_name: str = 'latitude',
               wavenumber_name: str = 'zonal_wavenumber', scale_by_circumference: bool = False):
    if which not in ('predictions', 'targets'):
      raise ValueError(f'Unhandled {which=}')
    self._which = which
    self._lon, self._lat, self._k = longitude_name, latitude_name, wavenumber_name
    self._circ = scale_by_circumference

  @property
  def unique_name(self) -> str:
    return f"Zonal{'Energy' if self._circ else 'Power'}Spectrum_{self._which}"

  def _compute_per_variable(self, predictions, targets):
    src = xr.as_dataarray(predictions if self._which == 'predictions' else targets)
    if self._lon not in src.dims:
      return None
    if self._circ and self._lat not in src.dims:
      raise ValueError(f'scale_by_circumference needs a {self._lat!r} dimension')
    return LazySpectrum(src, self._lon, self._k, self._circ, self._lat)


def ZonalEnergySpectrum(which: str = 'predictions', **kw):  # pylint: disable=invalid-name
  """S_k * 2 pi R cos(lat): zonal energy per unit wavenumber (WeatherBench-2 convention)."""
  return ZonalPowerSpectrum(which=which, scale_by_circumference=True, **kw)
