"""File-backed chunk loaders under the chunk feeder (counterpart of weatherbenchX/data_loaders/base.py:58-170 and
weatherbenchX/data_loaders/xarray_loaders.py:143-316, whose zarr / xarray backends are not installable here).

A loader maps (init_times, lead_times) to {variable: DataArray} exactly like the reference's `DataLoader.load_chunk`:
  * `PredictionsFromFiles`  fields indexed [init_time, lead_time, ...]: the chunk is the (init, lead) block
                            (PredictionsFromXarray, xarray_loaders.py:176-221);
  * `TargetsFromFiles`      fields indexed by ONE time axis: the chunk is gathered at valid_time = init_time + lead_time and
                            carries init_time / lead_time dims plus a 2-D `valid_time` coordinate
                            (TargetsFromXarray, xarray_loaders.py:224-264);
  * `ClimatologyFromFiles`  a [dayofyear, hour, ...] climatology read as predictions at the valid times' dayofyear / hour
                            (ClimatologyFromXarray, xarray_loaders.py:266-316);
  * `PersistenceFromFiles`  the analysis at init_time repeated along lead_time (PersistenceFromXarray, :319-337);
  * `ProbabilisticClimatologyFromFiles`  the same dayofyear / hour of `start_year..end_year` as ensemble members
                            (ProbabilisticClimatologyFromXarray, :340-409).
Storage: `.npy` files (numpy memory maps), NetCDF-3 (scipy.io.netcdf_file with mmap, the format io.py writes) or zarr v2
array directories (`ZarrArray`: read without the zarr package; uncompressed or zlib / gzip / bz2 / lzma chunks).  A chunk is
read -- decoded, if the file is big-endian -- STRAIGHT into page-locked memory (`pipeline.pinned_empty`), so the feeder's
upload is pure DMA on its copy stream (`wbx_memcpy_h2d_async`) and overlaps both the kernels of the previous chunk and this
loader's reads of the next one; the pool of page-locked blocks is the double buffer.  `add_nan_mask` attaches the `mask`
coordinate of data_loaders/base.py:25-56.  `timings` accumulates what the reads cost (bytes, seconds): disk / page-cache rate.

`device_layout='lon_fastest'`: real archives are stored [.., longitude, latitude] (xarray_loaders.py:185-188, 236-239 hand on the
store's order) and the zonal transforms are bound by L1 line requests on such rows (0.38 of the HBM peak against 0.60 on
longitude-fastest rows).  The copy into page-locked memory is then a blocked transposition of the last two dims
(`wbx_host_transpose`: 32 x 32 tiles, AVX2, split over the loader's threads), the chunk's DataArrays are [.., latitude,
longitude], and the DMA and every kernel see longitude-fastest fields; nothing is transposed on the device.
"""
from __future__ import annotations

import time
from typing import Mapping, Sequence

import numpy as np

from weatherbenchx_amd import xarray_lite as xr


class ZarrArray:
  """A zarr v2 array directory read WITHOUT the zarr package (not in the image): `.zarray` metadata + one file per chunk.
  The reference opens its archives with `xr.open_zarr` (data_loaders/xarray_loaders.py:143-175); this is the part of that the
  chunk feeder needs -- `shape`, `dtype`, `arr[i]` along the leading axes (lazy) and `read()` of what is left, assembled chunk by
  chunk.  Codecs: none, zlib, gzip, bz2, lzma (the standard library's; they release the GIL, so the loader's threads decode in
  parallel); blosc / zstd / lz4 (numcodecs) raise and say so.  Filters are not supported.  C and F chunk order, '.' and '/' chunk
  keys, edge chunks, missing chunks = fill_value."""

  def __init__(self, path: str, _meta=None, _prefix=()):
    import json  # pylint: disable=g-import-not-at-top
    import os  # pylint: disable=g-import-not-at-top
    self.path = path
    if _meta is None:
      meta_path = os.path.join(path, '.zarray')
      if not os.path.exists(meta_path):
        raise FileNotFoundError(f'{path}: no .zarray (zarr v2 array directories only; for a group pass (path, variable))')
      _meta = json.load(open(meta_path))
      if _meta.get('zarr_format') != 2:
        raise ValueError(f"{path}: zarr_format {_meta.get('zarr_format')}: only zarr v2")
      if _meta.get('filters'):
        raise ValueError(f'{path}: filters {_meta["filters"]} are not supported')
      comp = _meta.get('compressor')
      if comp is not None and comp.get('id') not in ('zlib', 'gzip', 'bz2', 'lzma'):
        raise ValueError(f"{path}: compressor {comp.get('id')!r} needs numcodecs, which this image does not have (supported: none, "
                         'zlib, gzip, bz2, lzma): re-chunk the store with one of those, or export the variable to .npy / NetCDF-3')
    self._meta, self._prefix = _meta, tuple(int(i) for i in _prefix)
    self._full_shape = tuple(int(n) for n in _meta['shape'])
    self._chunks = tuple(int(n) for n in _meta['chunks'])
    self.dtype = np.dtype(_meta['dtype'])
    self._sep = _meta.get('dimension_separator', '.')
    self._order = _meta.get('order', 'C')
    fill = _meta.get('fill_value')
    self._fill = np.nan if fill in ('NaN', None) and self.dtype.kind == 'f' else (0 if fill is None else fill)
    for i, n in zip(self._prefix, self._full_shape):
      if not 0 <= i < n:
        raise IndexError(f'index {i} out of range for an axis of {n}')

  @property
  def shape(self):
    return self._full_shape[len(self._prefix):]

  @property
  def ndim(self):
    return len(self.shape)

  @property
  def nbytes(self):
    return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

  def __getitem__(self, i):
    if isinstance(i, (int, np.integer)):
      if not self.shape:
        raise IndexError('too many indices')
      return ZarrArray(self.path, self._meta, self._prefix + (int(i) % self.shape[0] if i < 0 else int(i),))
    raise TypeError('ZarrArray: integer indices along the leading axes, then .read()')

  def _decode(self, raw: bytes) -> np.ndarray:
    comp = self._meta.get('compressor')
    if comp is not None:
      cid = comp['id']
      if cid == 'zlib':
        import zlib  # pylint: disable=g-import-not-at-top
        raw = zlib.decompress(raw)
      elif cid == 'gzip':
        import gzip  # pylint: disable=g-import-not-at-top
        raw = gzip.decompress(raw)
      elif cid == 'bz2':
        import bz2  # pylint: disable=g-import-not-at-top
        raw = bz2.decompress(raw)
      else:
        import lzma  # pylint: disable=g-import-not-at-top
        raw = lzma.decompress(raw)
    return np.frombuffer(raw, dtype=self.dtype).reshape(self._chunks, order=self._order)

  def read(self, out: np.ndarray | None = None) -> np.ndarray:
    """What is left after the leading indices, as one array in the file's dtype (`out`: written in place, any dtype numpy can
    cast to -- a big-endian store is decoded on the way)."""
    import itertools  # pylint: disable=g-import-not-at-top
    import os  # pylint: disable=g-import-not-at-top
    shape = self.shape
    if out is None:
      out = np.empty(shape, self.dtype.newbyteorder('='))
    elif tuple(out.shape) != tuple(shape):
      raise ValueError(f'out has shape {out.shape}, the selection {shape}')
    npre = len(self._prefix)
    lead = tuple(i // c for i, c in zip(self._prefix, self._chunks))
    within = tuple(i % c for i, c in zip(self._prefix, self._chunks))
    grid = [range(-(-n // c)) for n, c in zip(shape, self._chunks[npre:])]
    for idx in itertools.product(*grid):
      key = self._sep.join(str(i) for i in lead + idx) if lead + idx else '0'
      lo = [i * c for i, c in zip(idx, self._chunks[npre:])]
      hi = [min(l + c, n) for l, c, n in zip(lo, self._chunks[npre:], shape)]
      dst = out[tuple(slice(l, h) for l, h in zip(lo, hi))]
      f = os.path.join(self.path, *key.split('/')) if self._sep == '/' else os.path.join(self.path, key)
      if not os.path.exists(f):
        dst[...] = self._fill
        continue
      with open(f, 'rb') as fh:
        chunk = self._decode(fh.read())
      np.copyto(dst, chunk[within + tuple(slice(0, h - l) for l, h in zip(lo, hi))], casting='unsafe')
    return out

  def __array__(self, dtype=None, copy=None):  # pylint: disable=unused-argument
    a = self.read()
    return a if dtype is None else a.astype(dtype)


class _FileSource:
  """One variable in a file: an array-like that supports numpy fancy indexing along its leading axes."""

  def __init__(self, path: str, variable: str | None = None):
    self.path, self.variable = path, variable
    self._arr = None

  def array(self):
    if self._arr is None:
      import os  # pylint: disable=g-import-not-at-top
      if self.path.endswith('.npy'):
        self._arr = np.load(self.path, mmap_mode='r')
      elif os.path.isdir(self.path):  # a zarr v2 array directory, or a group + the variable's name
        self._arr = ZarrArray(os.path.join(self.path, self.variable) if self.variable else self.path)
      else:
        from scipy.io import netcdf_file  # pylint: disable=g-import-not-at-top
        self._file = netcdf_file(self.path, 'r', mmap=True)  # (kept open: the variable is a view of its map)
        self._arr = self._file.variables[self.variable].data
    return self._arr

  def __getstate__(self):  # loaders travel to worker processes (the reference pickles its DoFns): reopen there
    return {'path': self.path, 'variable': self.variable, '_arr': None}


def _strip_pool(state):
  state = dict(state)
  state['_pool'] = None
  return state


class FileLoader:
  """Base of the file-backed loaders.  `sources` = {variable: path.npy | (path.nc, name)}; `dims` = the stored dim order of
  every variable (after the time axes), e.g. ('level', 'longitude', 'latitude'); `coords` = {dim: values}."""

  def __init__(self, sources: Mapping[str, object], dims: Sequence[str], coords: Mapping[str, np.ndarray], *, add_nan_mask: bool = False,
               pinned: bool = True, threads: int = 4, device_layout: str | None = None):
    self._sources = {k: (_FileSource(v) if isinstance(v, str) else _FileSource(*v)) for k, v in sources.items()}
    self._stored_dims = tuple(dims)
    self._swap = _needs_swap(self._stored_dims, device_layout)
    # the dims of the chunks handed out: the stored order, or with the last two exchanged
    self._dims = self._stored_dims[:-2] + (self._stored_dims[-1], self._stored_dims[-2]) if self._swap else self._stored_dims
    self._coords = {k: np.asarray(v) for k, v in coords.items()}
    self._add_nan_mask = add_nan_mask
    self._pinned = pinned
    self._threads = max(1, int(threads))  # a chunk's gather is split over this many threads (numpy's copies release the GIL)
    self._pool = None
    self.timings = {'bytes': 0, 'seconds': 0.0, 'chunks': 0}

  def _buffer(self, shape, dtype):
    if self._pinned:
      from weatherbenchx_amd import pipeline  # pylint: disable=g-import-not-at-top
      try:
        return pipeline.pinned_empty(shape, dtype)
      except Exception:  # pylint: disable=broad-except  (no device in this process: plain memory, pageable upload later)
        pass
    return np.empty(shape, dtype)

  def _out_shape(self, shape):
    """The shape of a stored item as it is handed out."""
    shape = tuple(shape)
    return shape[:-2] + (shape[-1], shape[-2]) if self._swap else shape

  def _gather_swapped(self, arr, index, out):
    """out[a] = arr[index[a]] with the last two axes exchanged: the copy out of the page cache IS the layout change
    (`wbx_host_transpose`, blocked; planes dealt to the loader's threads -- ctypes releases the GIL)."""
    from weatherbenchx_amd import _hip  # pylint: disable=g-import-not-at-top
    lib = _hip.load_library()
    t0 = time.perf_counter()
    rows, cols = (int(n) for n in arr.shape[-2:])
    nb = int(np.prod(arr.shape[1:-2], dtype=np.int64))
    native = np.dtype(arr.dtype).newbyteorder('=')
    n = len(index)

    def part(a, b0, b1):
      src = arr[int(index[a])].reshape(nb, rows, cols)[b0:b1]
      if not (arr.dtype.isnative and src.flags['C_CONTIGUOUS']):
        src = np.ascontiguousarray(src, dtype=native)  # a big-endian file (NetCDF-3): decoded first
      dst = out[a].reshape(nb, cols, rows)[b0:b1]
      _hip.check(lib.wbx_host_transpose(dst.ctypes.data, src.ctypes.data, b1 - b0, rows, cols, native.itemsize), 'wbx_host_transpose')
    if native.itemsize not in (4, 8):
      raise TypeError(f'device_layout: fields of {arr.dtype} cannot be transposed by the loader (float32 / float64 only)')
    units = [(a, 0, nb) for a in range(n)]
    if self._threads > 1 and out.nbytes >= (8 << 20):
      per = max(1, -(-n * nb // (4 * self._threads)))  # ~4 units per thread
      units = [(a, b0, min(b0 + per, nb)) for a in range(n) for b0 in range(0, nb, per)]
      if self._pool is None:
        from concurrent.futures import ThreadPoolExecutor  # pylint: disable=g-import-not-at-top
        self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix='wbx-loader')
      list(self._pool.map(lambda u: part(*u), units))
    else:
      for u in units:
        part(*u)
    self.timings['seconds'] += time.perf_counter() - t0
    self.timings['bytes'] += out.nbytes

  def _gather_zarr(self, arr, index, out):
    """out[a] = arr[index[a]] out of a zarr store: every item is assembled from its chunk files (decompressed on the loader's
    threads) -- straight into `out`, or into a scratch array first when the last two dims are exchanged on the way."""
    t0 = time.perf_counter()
    native = np.dtype(arr.dtype).newbyteorder('=')

    def part(a):
      item = arr[int(index[a])]
      if not self._swap:
        item.read(out[a])
        return
      from weatherbenchx_amd import _hip  # pylint: disable=g-import-not-at-top
      tmp = item.read(np.empty(item.shape, native))
      rows, cols = (int(n) for n in tmp.shape[-2:])
      nb = int(np.prod(tmp.shape[:-2], dtype=np.int64))
      if native.itemsize not in (4, 8):
        raise TypeError(f'device_layout: fields of {arr.dtype} cannot be transposed by the loader (float32 / float64 only)')
      _hip.check(_hip.load_library().wbx_host_transpose(out[a].ctypes.data, tmp.ctypes.data, nb, rows, cols, native.itemsize),
                 'wbx_host_transpose')
    n = len(index)
    if self._threads > 1 and n > 1:
      if self._pool is None:
        from concurrent.futures import ThreadPoolExecutor  # pylint: disable=g-import-not-at-top
        self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix='wbx-loader')
      list(self._pool.map(part, range(n)))
    else:
      for a in range(n):
        part(a)
    self.timings['seconds'] += time.perf_counter() - t0
    self.timings['bytes'] += out.nbytes

  def _gather(self, arr, index, out):
    """out[...] = arr[index] along the leading axis, gathered STRAIGHT into `out` (no temporary), timed.  The indices have
    been validated (`_positions`), so `mode='clip'` only switches numpy's buffered copy off."""
    if isinstance(arr, ZarrArray):
      return self._gather_zarr(arr, index, out)
    if self._swap:
      return self._gather_swapped(arr, index, out)
    t0 = time.perf_counter()

    def part(lo, hi):
      if arr.dtype.isnative:
        np.take(arr, index[lo:hi], axis=0, out=out[lo:hi], mode='clip')
      else:  # a big-endian file (NetCDF-3): decoded on the way
        for a in range(lo, hi):
          np.copyto(out[a], arr[int(index[a])])
    n = len(index)
    # one memcpy thread moves ~24 GB/s out of the page cache (tools/bench_feeder_files.py), PCIe takes ~50: split the rows
    if self._threads > 1 and n > 1 and out.nbytes >= (8 << 20):
      if self._pool is None:
        from concurrent.futures import ThreadPoolExecutor  # pylint: disable=g-import-not-at-top
        self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix='wbx-loader')
      cuts = np.linspace(0, n, min(self._threads, n) + 1).astype(int)
      list(self._pool.map(lambda ab: part(*ab), zip(cuts[:-1], cuts[1:])))
    else:
      part(0, n)
    self.timings['seconds'] += time.perf_counter() - t0
    self.timings['bytes'] += out.nbytes

  def __getstate__(self):
    return _strip_pool(self.__dict__)

  def _finish(self, out):
    self.timings['chunks'] += 1
    if self._add_nan_mask:
      from weatherbenchx_amd import data as wdata  # pylint: disable=g-import-not-at-top
      out = wdata.add_nan_mask_to_data(out)
    return out


class PredictionsFromFiles(FileLoader):
  """Forecast fields stored [init_time, lead_time, *dims]."""

  def __init__(self, sources, init_times, lead_times, dims, coords, **kw):
    super().__init__(sources, dims, coords, **kw)
    self._init_times = np.asarray(init_times, dtype='datetime64[ns]')
    self._lead_times = np.asarray(lead_times, dtype='timedelta64[ns]')

  def load_chunk(self, init_times, lead_times=None):
    init_times = np.asarray(init_times, dtype='datetime64[ns]')
    ii = _positions(self._init_times, init_times, 'init_time')
    if lead_times is None:
      li = np.arange(self._lead_times.size)
    elif isinstance(lead_times, slice):
      # `.sel(lead_time=slice(start, stop))` (xarray_loaders.py:199-200): by LABEL, both bounds inclusive, None = open; the
      # step is not used (time_chunks.py:54-56)
      lo, hi = (None if b is None else np.asarray(b, dtype='timedelta64[ns]') for b in (lead_times.start, lead_times.stop))
      keep = np.ones(self._lead_times.size, dtype=bool)
      if lo is not None:
        keep &= self._lead_times >= lo
      if hi is not None:
        keep &= self._lead_times <= hi
      li = np.nonzero(keep)[0]
    else:
      li = _positions(self._lead_times, np.asarray(lead_times, dtype='timedelta64[ns]'), 'lead_time')
    out = {}
    for name, src in self._sources.items():
      arr = src.array()
      buf = self._buffer((ii.size, li.size) + self._out_shape(arr.shape[2:]), np.dtype(arr.dtype).newbyteorder('='))
      for a, i in enumerate(ii):
        self._gather(arr[int(i)], li, buf[a])
      coords = dict(self._coords, init_time=self._init_times[ii], lead_time=self._lead_times[li])
      out[name] = xr.DataArray(buf, dims=('init_time', 'lead_time') + self._dims, coords={k: v for k, v in coords.items() if k in ('init_time', 'lead_time') + self._dims},
                               name=name)
    return self._finish(out)


class TargetsFromFiles(FileLoader):
  """Analysis / observation fields stored [time, *dims]; a chunk is gathered at valid_time = init_time + lead_time."""

  def __init__(self, sources, times, dims, coords, **kw):
    super().__init__(sources, dims, coords, **kw)
    self._times = np.asarray(times, dtype='datetime64[ns]')

  def load_chunk(self, init_times, lead_times=None):
    init_times = np.asarray(init_times, dtype='datetime64[ns]')
    if isinstance(lead_times, slice):
      raise ValueError('Lead time slice not supported for target data loaders.')  # xarray_loaders.py:288-289
    if lead_times is None:  # the init times ARE the valid times (xarray_loaders.py:290-297)
      ti = _positions(self._times, init_times, 'valid_time')
      out = {}
      for name, src in self._sources.items():
        arr = src.array()
        buf = self._buffer((ti.size,) + self._out_shape(arr.shape[1:]), np.dtype(arr.dtype).newbyteorder('='))
        self._gather(arr, ti, buf)
        out[name] = xr.DataArray(buf, dims=('init_time',) + self._dims,
                                 coords={k: v for k, v in dict(self._coords, init_time=init_times).items() if k in ('init_time',) + self._dims}, name=name)
      return self._finish(out)
    lead_times = np.asarray(lead_times, dtype='timedelta64[ns]')
    valid = init_times[:, None] + lead_times[None, :]
    ti = _positions(self._times, valid.reshape(-1), 'valid_time')
    out = {}
    for name, src in self._sources.items():
      arr = src.array()
      shape = (init_times.size, lead_times.size) + self._out_shape(arr.shape[1:])
      buf = self._buffer(shape, np.dtype(arr.dtype).newbyteorder('='))
      self._gather(arr, ti, buf.reshape((ti.size,) + self._out_shape(arr.shape[1:])))
      coords = {k: v for k, v in dict(self._coords, init_time=init_times, lead_time=lead_times).items() if k in ('init_time', 'lead_time') + self._dims}
      da = xr.DataArray(buf, dims=('init_time', 'lead_time') + self._dims, coords=coords, name=name)
      out[name] = da.assign_coords(valid_time=xr.DataArray(valid, dims=('init_time', 'lead_time')))
    return self._finish(out)


class ClimatologyFromFiles(FileLoader):
  """A climatology stored [dayofyear, hour, *dims] (or [dayofyear, *dims] with `hours=False`) read as PREDICTIONS: the chunk
  of (init_times, lead_times) holds the climatology at dayofyear / hour of valid_time = init_time + lead_time, dims
  (init_time, lead_time, *dims) plus the 2-D `dayofyear` / `hour` coordinates the reference's vectorised `.sel` leaves --
  `ClimatologyFromXarray` (xarray_loaders.py:266-316).  Without lead times the init times are the valid times; a lead-time
  slice raises like the reference.  `dayofyear` / `hours`: the labels of the two leading axes (default 1..n and the
  24/n-hourly hours of the day)."""

  def __init__(self, sources, dims, coords, *, dayofyear=None, hours=None, **kw):
    super().__init__(sources, dims, coords, **kw)
    self._dayofyear = None if dayofyear is None else np.asarray(dayofyear)
    self._has_hour = hours is not False
    self._hours = None if hours is None or hours is False else np.asarray(hours)

  def _axes(self, arr):
    lead_axes = 2 if self._has_hour else 1
    if arr.ndim != lead_axes + len(self._stored_dims):
      raise ValueError(f'a climatology file holds [dayofyear, {"hour, " if self._has_hour else ""}*dims]: found {arr.ndim} axes for dims '
                       f'{self._stored_dims}')
    doy = self._dayofyear if self._dayofyear is not None else np.arange(1, arr.shape[0] + 1)
    if not self._has_hour:
      return doy, None
    if self._hours is not None:
      return doy, self._hours
    if 24 % arr.shape[1]:
      raise ValueError(f'{arr.shape[1]} hours of the day do not divide 24: pass hours=')
    return doy, np.arange(arr.shape[1]) * (24 // arr.shape[1])

  def load_chunk(self, init_times, lead_times=None):
    init_times = np.asarray(init_times, dtype='datetime64[ns]')
    if isinstance(lead_times, slice):
      raise ValueError('Lead time slice not yet supported for climatology data loaders.')  # xarray_loaders.py:305-308
    if lead_times is None:  # the init times ARE the valid times (xarray_loaders.py:309-314)
      lead, valid = None, init_times
      tdims, tshape = ('init_time',), (init_times.size,)
    else:
      lead = np.asarray(lead_times, dtype='timedelta64[ns]')
      valid = (init_times[:, None] + lead[None, :]).reshape(-1)
      tdims, tshape = ('init_time', 'lead_time'), (init_times.size, lead.size)
    day = valid.astype('datetime64[D]')
    doy_of = (day - valid.astype('datetime64[Y]').astype('datetime64[D]')).astype(np.int64) + 1
    hour_of = (valid - day).astype('timedelta64[h]').astype(np.int64)
    tcoords = {'init_time': init_times}
    if lead is not None:
      tcoords['lead_time'] = lead
    out = {}
    for name, src in self._sources.items():
      arr = src.array()
      doy, hours = self._axes(arr)
      di = _positions(doy, doy_of, 'dayofyear')
      item = self._out_shape(arr.shape[1 if hours is None else 2:])
      buf = self._buffer(tshape + item, np.dtype(arr.dtype).newbyteorder('='))
      flat = buf.reshape((valid.size,) + item)
      if hours is None:
        self._gather(arr, di, flat)
      else:
        hi = _positions(hours, hour_of, 'hour')
        # one day's hours are adjacent on disk: gather each day's run of valid times in one call
        order = np.argsort(di, kind='stable')
        if np.array_equal(order, np.arange(order.size)):  # (the common case: valid times ascending within a year)
          cuts = np.r_[0, np.nonzero(np.diff(di))[0] + 1, di.size]
          for lo, up in zip(cuts[:-1], cuts[1:]):
            self._gather(arr[int(di[lo])], hi[lo:up], flat[lo:up])
        else:
          for a in range(valid.size):
            self._gather(arr[int(di[a])], hi[a:a + 1], flat[a:a + 1])
      coords = {k: v for k, v in dict(self._coords, **tcoords).items() if k in tdims + self._dims}
      da = xr.DataArray(buf, dims=tdims + self._dims, coords=coords, name=name)
      extra = {'dayofyear': xr.DataArray(doy_of.reshape(tshape), dims=tdims)}
      if hours is not None:
        extra['hour'] = xr.DataArray(hour_of.reshape(tshape), dims=tdims)
      out[name] = da.assign_coords(extra)
    return self._finish(out)


class PersistenceFromFiles(TargetsFromFiles):
  """Analysis fields stored [time, *dims] read as a PERSISTENCE forecast: the field at init_time repeated along lead_time
  (`PersistenceFromXarray`, xarray_loaders.py:319-337).  Exact lead times are required."""

  def load_chunk(self, init_times, lead_times=None):
    if lead_times is None or isinstance(lead_times, slice):
      raise ValueError('Exact lead times must be specified for persistence data loader.')  # xarray_loaders.py:330-333
    init_times = np.asarray(init_times, dtype='datetime64[ns]')
    lead = np.asarray(lead_times, dtype='timedelta64[ns]')
    ti = _positions(self._times, init_times, 'valid_time')
    out = {}
    for name, src in self._sources.items():
      arr = src.array()
      item = self._out_shape(arr.shape[1:])
      buf = self._buffer((init_times.size, lead.size) + item, np.dtype(arr.dtype).newbyteorder('='))
      if lead.size:
        first = np.empty((init_times.size,) + item, buf.dtype) if lead.size > 1 else buf[:, 0]
        self._gather(arr, ti, first)  # read (and transposed) once ...
        buf[...] = first[:, None]     # ... then repeated along lead_time in page-locked memory
      coords = {k: v for k, v in dict(self._coords, init_time=init_times, lead_time=lead).items() if k in ('init_time', 'lead_time') + self._dims}
      out[name] = xr.DataArray(buf, dims=('init_time', 'lead_time') + self._dims, coords=coords, name=name)
    return self._finish(out)


class ProbabilisticClimatologyFromFiles(TargetsFromFiles):
  """Analysis fields stored [time, *dims] read as an ENSEMBLE forecast whose members are the years `start_year..end_year`
  (inclusive): member m at (init, lead) is the field at the valid time's dayofyear / hour in year start_year + m; day 366
  in a non-leap year is 1 January of the next (`ProbabilisticClimatologyFromXarray`, xarray_loaders.py:340-409).  Dims
  (ensemble_dim, init_time, lead_time, *dims) as the reference's vectorised `.sel` orders them, with its 3-D `valid_time`
  coordinate.  Exact lead times are required."""

  def __init__(self, sources, times, dims, coords, *, start_year: int, end_year: int, ensemble_dim: str = 'number', **kw):
    super().__init__(sources, times, dims, coords, **kw)
    self._years = np.arange(int(start_year), int(end_year) + 1)
    self._ensemble_dim = ensemble_dim

  def load_chunk(self, init_times, lead_times=None):
    if lead_times is None or isinstance(lead_times, slice):
      raise ValueError('Exact lead times must be specified for persistence data loader.')  # (sic) xarray_loaders.py:381-384
    init_times = np.asarray(init_times, dtype='datetime64[ns]')
    lead = np.asarray(lead_times, dtype='timedelta64[ns]')
    valid = init_times[:, None] + lead[None, :]
    day = valid.astype('datetime64[D]')
    doy0 = (day - valid.astype('datetime64[Y]').astype('datetime64[D]')).astype(np.int64)
    hod = (valid - day).astype('timedelta64[h]').astype(np.int64)
    starts = np.array([np.datetime64(str(y)) for y in self._years], dtype='datetime64[ns]')
    member_times = starts[:, None, None] + ((doy0 * 24 + hod) * np.timedelta64(1, 'h').astype('timedelta64[ns]'))[None]
    ti = _positions(self._times, member_times.reshape(-1), 'valid_time')
    tdims = (self._ensemble_dim, 'init_time', 'lead_time')
    out = {}
    for name, src in self._sources.items():
      arr = src.array()
      item = self._out_shape(arr.shape[1:])
      buf = self._buffer(member_times.shape + item, np.dtype(arr.dtype).newbyteorder('='))
      self._gather(arr, ti, buf.reshape((ti.size,) + item))
      coords = dict(self._coords, init_time=init_times, lead_time=lead, **{self._ensemble_dim: np.arange(self._years.size)})
      da = xr.DataArray(buf, dims=tdims + self._dims, coords={k: v for k, v in coords.items() if k in tdims + self._dims}, name=name)
      out[name] = da.assign_coords(valid_time=xr.DataArray(member_times, dims=tdims))
    return self._finish(out)


_LON_NAMES, _LAT_NAMES = ('longitude', 'lon'), ('latitude', 'lat')


def _needs_swap(dims, device_layout) -> bool:
  """Does `device_layout` ('lon_fastest' | 'lat_fastest' | None) ask for the last two stored dims to be exchanged?"""
  if device_layout is None:
    return False
  if device_layout not in ('lon_fastest', 'lat_fastest'):
    raise ValueError(f"device_layout {device_layout!r}: 'lon_fastest', 'lat_fastest' or None")
  want, other = (_LON_NAMES, _LAT_NAMES) if device_layout == 'lon_fastest' else (_LAT_NAMES, _LON_NAMES)
  if len(dims) >= 1 and dims[-1] in want:
    return False
  if len(dims) >= 2 and dims[-2] in want and dims[-1] in other:
    return True
  raise ValueError(f'device_layout={device_layout!r}: the stored dims {tuple(dims)} do not end in (latitude, longitude) in either order')


def _positions(axis: np.ndarray, wanted: np.ndarray, what: str) -> np.ndarray:
  """Exact label lookup (`.sel` of the reference): KeyError for a label the file does not hold."""
  order = np.argsort(axis, kind='stable')
  pos = np.searchsorted(axis[order], wanted)
  pos = np.clip(pos, 0, axis.size - 1)
  hit = order[pos]
  if not np.array_equal(axis[hit], wanted):
    missing = wanted[axis[hit] != wanted]
    raise KeyError(f'{what} labels not in the file: {missing[:4]}')
  return hit


def load_chunk_fn(predictions: FileLoader, targets: FileLoader):
  """(init_chunk, lead_chunk) -> (predictions, targets) for pipeline.evaluate_chunks / evaluate_passes: the two
  `load_chunk` calls of LoadPredictionsAndTargets (beam_pipeline.py:69-116)."""
  def load(init_chunk, lead_chunk):
    return predictions.load_chunk(init_chunk, lead_chunk), targets.load_chunk(init_chunk, lead_chunk)
  load.loaders = (predictions, targets)
  return load
