"""Zonal power / energy spectra: fused in-LDS FFT + |F|^2 reduction (rocFFT for odd / strided rows).

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
from weatherbenchx_amd import replay
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base

EARTH_RADIUS_M = 6371.0e3


class _BatchGeometry:
  """How the rows of a field (all dims but longitude) map onto uniformly strided rocFFT batches."""

  def __init__(self, dims, sizes_by_dim, layout, lon_dim):
    row_dims = [d for d in dims if d != lon_dim]
    sizes = [sizes_by_dim[d] for d in row_dims]
    strides = [layout.stride(d) for d in row_dims]
    self.lon_stride = layout.stride(lon_dim)
    order = np.argsort(strides)[::-1] if row_dims else []
    # the largest suffix (in stride order) that nests uniformly: stride[i] == stride[i+1] * size[i+1]
    inner = []
    for idx in order[::-1]:
      if not inner or strides[idx] == strides[inner[-1]] * sizes[inner[-1]]:
        inner.append(idx)
      else:
        break
    inner_set = set(inner)
    outer = [i for i in range(len(row_dims)) if i not in inner_set]
    self.batch = int(np.prod([sizes[i] for i in inner], dtype=np.int64)) if inner else 1
    self.row_stride = strides[inner[0]] if inner else 1
    nrows = int(np.prod(sizes, dtype=np.int64)) if sizes else 1
    row_index = np.arange(nrows, dtype=np.int64).reshape(sizes or [1])
    perm = outer + sorted(inner, key=lambda i: -strides[i])
    self.row_index = np.transpose(row_index, perm).reshape(-1, self.batch) if row_dims else row_index.reshape(1, 1)
    off = np.zeros((), dtype=np.int64)
    for i in outer:
      off = off[..., None] + np.arange(sizes[i], dtype=np.int64) * strides[i]
    self.outer_offsets = np.asarray(off).reshape(-1)
    # rows walked in C order by every slab?  then group/scale can be addressed by pointer offset
    self.c_order = all(bool(np.all(np.diff(r) == 1)) for r in self.row_index) if self.batch > 1 else True
    self.permutation = None if self.c_order else self.row_index.reshape(-1)


_geometry_cache: dict = {}


def _geometry(field: xr.DataArray, layout, lon_dim) -> _BatchGeometry:
  key = (field.dims, field.shape, tuple(sorted(layout.strides.items(), key=str)), lon_dim)
  if key not in _geometry_cache:
    if len(_geometry_cache) > 32:
      _geometry_cache.clear()
    _geometry_cache[key] = _BatchGeometry(field.dims, field.sizes, layout, lon_dim)
  return _geometry_cache[key]


def _run_spectrum(field: xr.DataArray, lon_dim: str, group: np.ndarray, scale: np.ndarray, ngroup: int,
                  cache: dict | None = None) -> np.ndarray:
  """power[ngroup][nk]; rows = all non-longitude dims in the field's own order.  `cache` (owned by the caller,
  e.g. the aggregator's weight object) keeps the uploaded group/scale arrays between chunks."""
  ctx = _hip.default_context()
  fused = field.__dict__.pop('_wbx_fused_spectrum', None)
  if fused is not None and cache is not None and fused['cache'] is cache and fused['ngroup'] == ngroup and fused['ctx'] is ctx:
    # engine.fuse_det_spectra: the deterministic launch over this very field already produced its spectrum (wbx_det_spectrum)
    return engine._deliver(ctx, fused['ptr'], (ngroup, field.sizes[lon_dim] // 2 + 1))  # pylint: disable=protected-access
  data = field.data
  engine._sync_torch_producers([data])  # pylint: disable=protected-access
  if engine._common_dtype([data]) != _hip.F32:  # pylint: disable=protected-access
    raise TypeError('zonal spectra take float32 fields (rocFFT single precision); cast the input')
  dev = engine._to_device(ctx, field, _hip.F32)  # pylint: disable=protected-access
  engine._order_uploads(ctx, [dev])  # pylint: disable=protected-access
  nlon = field.sizes[lon_dim]
  nk = nlon // 2 + 1
  geo = _geometry(field, dev.layout, lon_dim)
  ckey = (id(geo), ngroup)
  bufs = cache.get(ckey) if cache is not None else None
  if bufs is None:
    g = np.ascontiguousarray(group, dtype=np.int32)
    sc = np.ascontiguousarray(scale, dtype=np.float64)
    if geo.permutation is not None:  # slabs walk the rows in another order than C order
      g, sc = g[geo.permutation], sc[geo.permutation]
    offs = np.ascontiguousarray(geo.outer_offsets, dtype=np.int64)
    if offs.size > 1 and g.size == offs.size * geo.batch:
      # the slabs of one output group next to each other (a stable sort on each slab's first group): the kernels keep their
      # fp64 sums in registers / LDS while the group stays, and [lead, level] slabs in storage order change it every slab
      order = np.argsort(g.reshape(offs.size, geo.batch)[:, 0], kind='stable')
      if np.any(order != np.arange(offs.size)):
        offs = np.ascontiguousarray(offs[order])
        g = np.ascontiguousarray(g.reshape(offs.size, geo.batch)[order].reshape(-1))
        sc = np.ascontiguousarray(sc.reshape(offs.size, geo.batch)[order].reshape(-1))
    bufs = (ctx.upload(g), ctx.upload(sc), geo, offs)
    if cache is not None:
      cache[ckey] = bufs
  g_dev, s_dev = bufs[0], bufs[1]
  replay.keep(bufs)  # (a chunk that is being recorded: the row tables -- device and host -- belong to the record)
  out = engine._scratch(ctx, 'spectrum', max(ngroup * nk, 1) * 8)  # pylint: disable=protected-access
  # one call for every slab (lead x level slabs of adjacent latitude rows for latitude-fastest fields); group / scale are
  # already in slab-major row order (geo.permutation)
  offs = bufs[3]

  def call():
    _hip.check(ctx.lib.wbx_zonal_spectrum_slabs(ctx.handle, C.c_void_p(dev.ptr), int(geo.lon_stride), int(geo.row_stride),
                                                int(geo.batch), int(offs.size), offs.ctypes.data_as(C.c_void_p), int(nlon),
                                                C.c_void_p(g_dev.ptr), C.c_void_p(s_dev.ptr), int(ngroup), 0,
                                                C.c_void_p(out.ptr)), 'wbx_zonal_spectrum_slabs')
  engine.timed_launch(ctx, call, kind='spectrum', rows=int(offs.size * geo.batch), nlon=int(nlon), lon_stride=int(geo.lon_stride))
  return engine._deliver(ctx, out.ptr, (ngroup, nk))  # pylint: disable=protected-access


class LazySpectrum(xr.LazyPickleMixin, xr.DataArray):
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

  def rows_entry(self, row_weight: xr.DataArray, kept_dims):
    """-> (entry, ngroup, kept, sizes): the per-row output group and scale of `sum over the non-kept row dims of weight *
    spectrum` -- data independent, built once per (weight object, frame) and kept with the weight object (`entry['dev']`
    holds the device copies)."""
    row_dims = [d for d in self._source.dims if d != self._lon_dim]
    sizes = {d: self._source.sizes[d] for d in row_dims}
    kept = [d for d in row_dims if d in kept_dims]
    ngroup = int(np.prod([sizes[d] for d in kept], dtype=np.int64)) if kept else 1
    store = row_weight.__dict__.setdefault('_wbx_spectrum', {})
    fkey = (tuple(row_dims), tuple(sizes[d] for d in row_dims), tuple(kept), self._circumference)
    entry = store.get(fkey)
    if entry is None:
      shape = [sizes[d] for d in row_dims]
      w = (row_weight * self.row_scale()).astype(np.float64)
      scale = np.broadcast_to(xr._bcast_data(w, row_dims, sizes), shape).reshape(-1)  # pylint: disable=protected-access
      g = np.zeros((), dtype=np.int64)
      mult, mults = 1, {}
      for d in reversed(kept):
        mults[d] = mult
        mult *= sizes[d]
      for d in row_dims:
        g = g[..., None] + np.arange(sizes[d], dtype=np.int64) * mults.get(d, 0)
      entry = {'group': np.broadcast_to(g, shape).reshape(-1), 'scale': scale, 'dev': {}, 'row_dims': tuple(row_dims),
               'row_shape': tuple(shape)}
      if len(store) > 8:
        store.clear()
      store[fkey] = entry
    return entry, ngroup, kept, sizes

  def reduce_rows(self, row_weight: xr.DataArray, kept_dims) -> np.ndarray:
    """sum over the non-kept row dims of weight * spectrum -> array over (kept_dims..., k)."""
    entry, ngroup, kept, sizes = self.rows_entry(row_weight, kept_dims)
    out = _run_spectrum(self._source, self._lon_dim, entry['group'], entry['scale'], ngroup, cache=entry['dev'])
    return out.reshape([sizes[d] for d in kept] + [self._nk]), tuple(kept) + (self._k_dim,)

  @property
  def data(self):
    if self._data is None:
      row_dims = [d for d in self._source.dims if d != self._lon_dim]
      with engine.synchronous_results():
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
